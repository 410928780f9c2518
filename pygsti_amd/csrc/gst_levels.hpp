// gst_levels.hpp -- log-depth evaluation of the state tries (second half / H2-iii).
//
// The walk programs of gst_plan.hpp evaluate a task's trie as ONE dependent chain of mat-vecs: a GST family
// (preparation fiducial . germ^p . measurement fiducial, p up to 1,024) is ~1,150 strictly sequential steps, 0.5 ms of
// pure latency whatever the hardware.  That order is part of the FD mode's bit-parity contract and stays there.  The
// modes WITHOUT an ordering contract -- exact derivatives (Matrix-simulator semantics, <= 1e-8) and the opt-in fast
// probabilities (<= 1e-10) -- only need every state of the trie to within rounding, and a germ-power path has a
// log-depth evaluation (the reference's Matrix path multiplies sub-products over an eval tree for the same reason:
// matrixforwardsim.py:675-727, evaltree.py:31-189):
//
//     F[r m]   = (G_germ)^r F[0]                      X_{2^k + j} = (G_germ)^(2^k) X_j   (doubling, k = 0 .. log2 q)
//     F[r m+s] = G_{g[s-1]} F[r m + s - 1]            for ALL r at once                    (m - 1 further levels)
//
// with (G_germ)^(2^k) from k squarings.  `build_level_program` finds the periodic paths of every task's trie (any period
// up to 16; nothing is assumed about fiducials, germs or powers), schedules the task as a short sequence of STAGES --
// each a set of independent 16-row tiles OUT = IN x M (rows = states, M = a gate, a germ power, or the identity) -- and
// the device executes a stage per barrier (gst_kernels_levels.hip: one workgroup per task, every tile four
// v_mfma_f64_16x16x4_f64).  A 2Q family's 1,150 steps become ~2 m + 15 stages.
//
// Conventions.  A state is stored [component][vector] with `nv` vectors per trie node (forward cache: nv = 1; backward
// cache of the reversed plan: nv = n_effects, cache[id][component][effect]).  Everything is written in ROW form:
// out_row = in_row x M_B, where M_B of a gate is G^T for the forward pass and G for the backward pass (the caller hands
// the table over in that layout); the product along a path is then M_B(g0) M_B(g1) ... in application order.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "gst_plan.hpp"

namespace gst {

constexpr int32_t LV_KIND_ROWS = 0;   // rows = trie states (ids pool): out[dst ids] = in[src ids] x M
constexpr int32_t LV_KIND_MAT = 1;    // rows = the 16 rows of a scratch matrix: slot[dst] = (slot or gate)[src] x M
constexpr int32_t LV_BMAT_IDENT = -1; // M = identity (copies start vectors into their trie node)
inline int32_t lv_slot_ref(int32_t slot) { return -(slot + 2); }          // matrix references: >= 0 gate, -1 identity, <= -2 scratch slot
inline int32_t lv_ref_slot(int32_t ref) { return -(ref + 2); }

struct LevelProgram {
    int32_t D = 0, nv = 1;
    // task t: words[task_off[t] ..): n_stages, then per stage: n_tiles, then 4 words per tile
    //   w0 = kind | (n_nodes << 8)   (ROWS: trie nodes in the tile, nv rows each, n_nodes * nv <= 16; MAT: 16)
    //   w1 = matrix reference of M
    //   w2 = ROWS: offset of the n_nodes source ids in `ids` (id >= 0: state id; id < 0: start vector -(id + 1))
    //        MAT : matrix reference of the source rows (gate or scratch slot)
    //   w3 = ROWS: offset of the n_nodes destination ids in `ids`;  MAT: destination scratch slot
    std::vector<int64_t> task_off;
    std::vector<int32_t> words;
    std::vector<int32_t> ids;
    std::vector<int64_t> task_ids_off;   // [n_tasks + 1]: task t's ids are ids[task_ids_off[t] .. task_ids_off[t + 1]) (tile offsets stay global)
    int64_t max_task_ints = 0;           // most words + ids of one task (what the kernel stages in LDS)
    int32_t max_mats = 0;          // scratch matrices a task needs at most (every slot is written ONCE per pass)
    int32_t max_stages = 0;
    int64_t n_stages = 0, n_tiles = 0, n_chains = 0, chain_nodes = 0, n_nodes = 0, n_produced = 0;
    int64_t sum_task_depth = 0;    // sum over tasks of the deepest node: the dependent steps of the sequential walk
    bool worthwhile = false;       // the stages are few against those steps
};

// Schedules every task of P (P.task_node0 / task_nodes / node_parent / node_sym).  nv * k == 16 for some k is required.
// Returns "" or why the plan has no level program (the caller then keeps the sequential walk).
// keep_ids (may be NULL = every state): produce only these states and the ones they are computed from -- with the circuits'
// final states this is the probability-only program (most positions inside a germ period are then never formed).
std::string build_level_program(const HostPlan& P, int32_t nv, LevelProgram& out, const std::vector<int32_t>* keep_ids = nullptr);

}  // namespace gst
