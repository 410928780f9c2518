// gst_plan.hpp -- host-side plan compiler: circuits -> prefix trie -> walk programs.
//
// Replaces, for the device path, what the reference does in PrefixTable (pygsti/layouts/prefixtable.py:26-101,
// 680-741: sort by length, cache only prefixes that are themselves circuits) and convert_maplayout
// (pygsti/forwardsims/mapforwardsim_calc_densitymx.pyx:55-77).  Design differs on purpose:
//   * sharing is over the FULL prefix trie (every distinct prefix is one state), not only over
//     prefixes that happen to be circuits;
//   * the trie is cut into independent "tasks" (contiguous runs of lexicographically sorted
//     circuits); each task is compiled to a linear walk program (RHO/APPLY/SAVE/LOAD/EMIT) that one
//     wavefront interprets, with save slots assigned statically (no run-time stack pointer).
// Every state is still rho followed by the circuit's gates applied left to right, so results are
// bit-identical to the reference's table walk whatever the sharing structure.
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace gst {

struct HostPlan {
    int32_t D = 0, n_gates = 0, n_rhos = 0, n_effects = 0;
    int64_t n_circuits = 0, n_elements = 0;

    // expanded circuits, CSR
    std::vector<int32_t> circ_rho;
    std::vector<int64_t> circ_ptr;
    std::vector<int32_t> circ_gates;
    // effect CSR indexed by circuit
    std::vector<int32_t> eff_ptr;   // int32 offsets on device
    std::vector<int32_t> eff_label, eff_dest;

    // compiled programs
    std::vector<uint32_t> prog;
    std::vector<int64_t> task_off;      // [n_tasks+1]
    std::vector<int64_t> task_applies;  // gate applications per task (scheduling weight)
    int32_t max_slots = 0, max_depth = 0;
    int64_t sum_depth = 0, trie_nodes = 0, applies_per_pass = 0;
    int64_t n_state_ids = 0;   // NODE marker ids: one per task-trie node (size of the base-state cache)
    std::vector<int32_t> node_parent, node_sym;   // [n_state_ids]: parent state id (-1: a rho state) and gate/rho index
    std::vector<int32_t> circ_leaf;               // [n_circuits]: state id of each circuit's final state
    std::vector<int64_t> task_node0;              // [n_tasks + 1]: the state ids of task t are task_node0[t] ... + task_nodes[t] - 1
    std::vector<int32_t> task_nodes;              //   (its first id is the task's virtual root: no state of its own)

    int64_t n_tasks() const { return (int64_t)task_off.size() - 1; }
};

// Fill circ_* from the reference's prefix-table format.  Returns "" or an error message.
std::string expand_table(HostPlan& P, int32_t n_rows, int32_t cache_size, const int32_t* t_dest,
                         const int32_t* t_start, const int32_t* t_cache, const int32_t* t_rho,
                         const int64_t* row_ptr, const int32_t* gate_idx);

// Validate indices, then build trie/tasks/programs.  target_tasks <= 0 picks a default; max_slots > 0
// caps the number of save slots a program may use (states beyond it are re-derived by replay).
std::string compile_plan(HostPlan& P, int32_t target_tasks, int32_t max_slots);

// Work estimate of a derivative pass per (task, perturbed object): cost[t * (n_gates + 2) + g] = gate applications the
// walk of task t executes when only gate g is perturbed (states before g's first use are taken from the base-state
// cache, saved clean states come back clean); column n_gates = a perturbed rho (every application); column
// n_gates + 1 = the task's EMIT count (what an effect-only wavefront still does).  Used to launch expensive
// (task, wavefront) pairs first.  Empty result when n_gates > 64.
void task_gate_costs(const HostPlan& P, std::vector<int32_t>& cost);

// EXACT work of one finite-difference pass of the lane-per-model walk (gst_kernels.hip) for a given set of wavefronts, by
// the kernel's own clean/dirty rule: wavefront w computes a gate application only once a gate one of its lanes perturbs
// (wave_gates[w], bit per gate) has been applied on the path -- or from the start when a lane perturbs a preparation
// (wave_rho[w]); SAVE / LOAD carry the flag; an EMIT costs real dot products when the state is dirty or a lane perturbs an
// effect (wave_eff[w]), otherwise it stores exact zeros.  wave_cols[w] = live lanes of w (<= 64).
// out[0] = wavefront-applications executed, out[1] = column-applications executed (live lanes only),
// out[2] = wavefront-dots executed, out[3] = column-dots executed, out[4] = wavefront-applications of the whole schedule
// (n_waves * applies_per_pass: what a kernel without the rule would execute), out[5] = column-applications of the schedule.
void fd_executed_work(const HostPlan& P, const std::vector<uint64_t>& wave_gates, const std::vector<uint8_t>& wave_rho,
                      const std::vector<uint8_t>& wave_eff, const std::vector<int32_t>& wave_cols, int64_t out[6]);

// Hand-over points of the derivative walks (persistent launch of small atoms): for every task a program position
// (word index relative to the task's first word, or -1) at which NO save slot is live, so that a walk can be cut there
// into two halves that run on different SIMDs -- the first stores its 64 lane states, the second picks them up -- and
// the fraction of the task's gate applications that lie before it.  The position closest to the middle is chosen; it
// is always an op the interpreter's outer loop sees (the word after a LOAD, or an EMIT).
void task_split_points(const HostPlan& P, std::vector<int32_t>& split_pc, std::vector<float>& split_frac);
// Every position of a task's program at which a walk can be handed from one wavefront to another -- no save slot is
// live, the 64 lane states are the walk's entire state -- between 10 % and 90 % of its gate applications, thinned to
// max_per_task evenly spaced ones: cand_pc[cand_ptr[t] .. cand_ptr[t + 1]) (word index relative to the task start,
// ascending) with cand_frac = fraction of the task's gate applications in front of it.
// "Dirty programs" (finite differences over whole-object perturbations, gst_kernels_pert.hip): for every task and every
// object class c -- gate c for c < n_gates, state preparation c - n_gates after that -- the walk program with everything
// removed that is bit-identical to the base pass when ONLY that object is perturbed.  What remains starts each perturbed
// excursion with `CACHE id` (v <- state `id` of the base pass, GST_OP_CACHE) or the perturbed `RHO`, then APPLY / SAVE /
// LOAD / EMIT as in the task's program (no NODE markers: nothing is cached); EMITs reached in a clean state are dropped
// (their Jacobian entries are exact zeros).  Program of (task t, class c): words[off[t * n_classes + c] .. off[.. + 1]),
// empty when the task never shows the object to an outcome.  applies / emits: its gate applications / EMITs.
struct DirtyPrograms {
    int32_t n_classes = 0;
    std::vector<uint32_t> words;
    std::vector<int64_t> off;
    std::vector<int32_t> applies, emits;
};
void build_dirty_programs(const HostPlan& P, DirtyPrograms& out);

// Per-SIMD queues of the persistent FD launch.  `items` = (-estimated cost, pair id = task * n_units + unit), longest
// first.  Longest-processing-time packing into n_bins queues, then HAND-OVERS: a walk of a queue above the mean is cut
// at the slot-free position (task_split_candidates) that brings its queue down to the mean; the first part stays at the
// FRONT of its queue, the second part goes to the emptiest queue, in cost order behind that queue's longer items (its
// wavefront waits for the first part's states).  handover: 0 = never cut, 1 = cut to balance, 2 = cut every walk that
// can be cut, in the middle (tests).  bin_items carry the part in their top two bits (0 whole, 1 first, 2 second).
struct FdQueues {
    std::vector<int32_t> bin_ptr;        // [n_bins + 1]
    std::vector<uint32_t> bin_items;
    std::vector<int64_t> load;           // estimated work per queue
    int32_t n_split = 0;
    std::vector<int32_t> ho_index;       // [n_tasks * n_units]: hand-over buffer of a cut pair, or -1 (empty when n_split == 0)
    std::vector<int32_t> ho_pc;          // [n_split]: where the pair is cut (word index relative to its task's start)
    std::vector<uint32_t> ho_live;       // [n_split]: save slots that are live at the cut (bit s): they travel with the states
};
void pack_fd_queues(const std::vector<std::pair<int32_t, uint32_t>>& items, int32_t n_units, int64_t n_tasks, int n_bins, int handover,
                    const std::vector<int32_t>& cand_ptr, const std::vector<int32_t>& cand_pc, const std::vector<float>& cand_frac,
                    const std::vector<uint32_t>& cand_live, FdQueues& out);
// cand_live (may be NULL): also positions at which save slots ARE live (slots 0..3 only) qualify, and the mask of the
// live slots is returned per candidate -- the hand-over then carries those slots along with the lane states.
void task_split_candidates(const HostPlan& P, std::vector<int32_t>& cand_ptr, std::vector<int32_t>& cand_pc, std::vector<float>& cand_frac,
                           int max_per_task, std::vector<uint32_t>* cand_live);

// Analytic derivatives (gst_kernels_analytic.hip, MFMA path): the plan of the REVERSED circuits (one dummy start, the
// gates of every circuit in reverse order, no effects).  Walking it with the transposed gates from every effect
// vector gives the backward states B_k = (G_n ... G_{k+1})^T E shared over common SUFFIXES, exactly as the forward
// walk shares prefixes.  Returns "" or an error message.
std::string build_reverse_plan(const HostPlan& P, HostPlan& R, int32_t target_tasks, int32_t max_slots);

// For every circuit c = g_1 ... g_n and every gate g: the gate applications k with g_k = g, as pairs
// (state id of F_{k-1} in P, state id of B_k in R), ascending k.  pos_ptr[c * n_gates + g] .. [c * n_gates + g + 1]
// delimit the pairs of (c, g) in pf / pr.  rev_leaf_state[c] = state id in R of B_0 (the whole reversed circuit).
void build_pair_tables(const HostPlan& P, const HostPlan& R, std::vector<int32_t>& pf, std::vector<int32_t>& pr,
                       std::vector<int64_t>& pos_ptr);

}  // namespace gst
