/*
 * oracle/ref_driver.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A thin driver (our code) around the REFERENCE's own C++ representation classes, compiled from
 * the sources where they lie under /root/reference (see oracle/Makefile, target _ref):
 *     pygsti/evotypes/densitymx/statecreps.cpp   (StateCRep)
 *     pygsti/evotypes/densitymx/opcreps.cpp      (OpCRep_Dense::acton, :40-54)
 *     pygsti/evotypes/densitymx/effectcreps.cpp  (EffectCRep_Dense::probability, :39-45)
 * The Cython layer that normally drives those classes (mapforwardsim_calc_densitymx.pyx) needs
 * Cython-generated code and is therefore "unbuildable here" by the task's rules; this file
 * re-expresses only its loop (pyx:194-287, pointer juggling included) and its FD wrapper
 * (pyx:349-381) and lets the reference's compiled arithmetic do the work.  The resulting
 * oracle/_ref/libgst_ref.so is used (a) to validate oracle/mapfill_oracle.c bit for bit and
 * (b) as bench.py's cpu_baseline with kind "reference".
 *
 * Outputs only into oracle/_ref/ (git-ignored); no reference source is copied into this repo.
 */
#include <cstdint>
#include <cstring>
#include <vector>

#include "statecreps.h"
#include "opcreps.h"
#include "effectcreps.h"

extern "C" {
#include "oracle_abi.h"
}

using namespace CReps_densitymx;

namespace {

struct RefModel {
    std::vector<double> g, r, e;
    std::vector<OpCRep*> ops;
    std::vector<StateCRep*> rhos;
    std::vector<EffectCRep*> effs;
    int D;
    RefModel(const oracle_plan* P, const oracle_model* M) : D(P->D) {
        g.assign(M->gates, M->gates + (size_t)M->nG * D * D);
        r.assign(M->rhos, M->rhos + (size_t)M->nR * D);
        e.assign(M->effects, M->effects + (size_t)M->nEl * D);
        for (int i = 0; i < M->nG; i++) ops.push_back(new OpCRep_Dense(g.data() + (size_t)i * D * D, D));
        for (int i = 0; i < M->nR; i++) rhos.push_back(new StateCRep(r.data() + (size_t)i * D, D, false));
        for (int i = 0; i < M->nEl; i++) effs.push_back(new EffectCRep_Dense(e.data() + (size_t)i * D, D));
    }
    ~RefModel() {
        for (auto p : ops) delete p;
        for (auto p : rhos) delete p;
        for (auto p : effs) delete p;
    }
    double* slot(const oracle_model* M, int p) {
        static double unused;   /* kind -1: object never applied by this atom */
        switch (M->pkind[p]) {
        case KIND_GATE:   return g.data() + (size_t)M->pobj[p] * D * D + M->pelem[p];
        case KIND_RHO:    return r.data() + (size_t)M->pobj[p] * D + M->pelem[p];
        case KIND_EFFECT: return e.data() + (size_t)M->pobj[p] * D + M->pelem[p];
        default:          return &unused;
        }
    }
};

/* the loop of dm_mapfill_probs (pyx:224-283) with the reference's pointer ownership rules */
void ref_probs(const oracle_plan* P, RefModel& R, std::vector<StateCRep*>& cache, double* out)
{
    StateCRep* prop2 = new StateCRep(P->D);
    StateCRep* shelved = new StateCRep(P->D);
    for (int k = 0; k < P->n_rows; k++) {
        const int i = P->t_dest[k], istart = P->t_start[k], icache = P->t_cache[k];
        StateCRep* init_state = (istart == -1) ? R.rhos[P->t_rho[k]] : cache[istart];
        StateCRep* prop1 = (icache == -1) ? shelved : cache[icache];
        prop1->copy_from(init_state);
        for (int64_t l = P->row_ptr[k]; l < P->row_ptr[k + 1]; l++) {
            R.ops[P->gate_idx[l]]->acton(prop1, prop2);
            StateCRep* t = prop1; prop1 = prop2; prop2 = t;
        }
        StateCRep* final_state = prop1;
        INT precomp_id = 0;
        for (int64_t e = P->eff_ptr[i]; e < P->eff_ptr[i + 1]; e++)
            out[P->eff_dest[e]] = R.effs[P->eff_label[e]]->probability_using_cache(final_state, prop2, precomp_id);
        if (icache != -1) cache[icache] = final_state;
        else shelved = final_state;
    }
    delete prop2;
    delete shelved;
}

std::vector<StateCRep*> make_cache(const oracle_plan* P)
{
    std::vector<StateCRep*> c(P->cache_size);
    for (auto& s : c) s = new StateCRep(P->D);
    return c;
}
void free_cache(std::vector<StateCRep*>& c) { for (auto s : c) delete s; }

}  // namespace

extern "C" int ref_fill_probs(const oracle_plan* P, const oracle_model* M, double* out)
{
    RefModel R(P, M);
    auto cache = make_cache(P);
    ref_probs(P, R, cache, out);
    free_cache(cache);
    return 0;
}

extern "C" int ref_fill_dprobs(const oracle_plan* P, const oracle_model* M, int64_t nE,
                               const int64_t* param_idx, const int64_t* dest_idx, int64_t n_param,
                               double eps, double* out, int64_t ld, double* probs_out)
{
    RefModel R(P, M);
    auto cache = make_cache(P);
    std::vector<double> probs(nE), probs2(nE);
    ref_probs(P, R, cache, probs.data());
    for (int64_t c = 0; c < n_param; c++) {
        double* s = R.slot(M, (int)param_idx[c]);
        const double orig = *s;
        *s = orig + eps;
        ref_probs(P, R, cache, probs2.data());
        const int64_t col = dest_idx ? dest_idx[c] : c;
        for (int64_t k = 0; k < nE; k++) out[k * ld + col] = (probs2[k] - probs[k]) / eps;
        *s = orig;
    }
    if (probs_out) std::memcpy(probs_out, probs.data(), sizeof(double) * nE);
    free_cache(cache);
    return 0;
}

extern "C" int ref_time_passes(const oracle_plan* P, const oracle_model* M, int64_t, int n_pass, double* out)
{
    RefModel R(P, M);
    auto cache = make_cache(P);
    for (int i = 0; i < n_pass; i++) ref_probs(P, R, cache, out);
    free_cache(cache);
    return 0;
}
