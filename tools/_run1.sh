cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
VARS="GST_FD_CUT_FRAC=0.55|GST_FD_CUT_FRAC=0.6|GST_FD_CUT_FRAC=0.65|GST_FD_CUT=2 GST_FD_CUT_GAIN=150|GST_FD_CUT=2 GST_FD_CUT_GAIN=300" EMU="8" timeout 900 bash tools/ab_env.sh 2>&1 | tee gpurun_out/r03_cut_ab2.txt
