"""pygsti_amd -- MI355X-native dense-matrix forward simulation (probs / dprobs / hprobs) for GST.

Only the hot path of pyGSTi's forward simulators is implemented here, behind the reference's own
method names; see DESIGN.md.  The compute lives in libgstfwd.so (HIP, gfx950); importing this
package does not need a GPU, calling a fill does.
"""
__version__ = "0.1.0"
