// Batched MCTS for MI355X (gfx950): B independent search trees advance in lock step, one
// wavefront per tree per phase.  Restates the semantics of the reference's single-tree agent
// (librubiks/solving/agents.py:415-645) so that every tree is node-for-node what the reference
// would build for that scramble given the same network outputs.
//
// The work per tree per iteration is a dozen rows, so these kernels are latency-bound, not
// bandwidth-bound: the design goal is few dependent memory round trips per phase, 12 lanes of the
// wave doing the per-action work in parallel (ballots / DPP reductions instead of loops), and
// no atomics across trees (each tree owns its node arrays and its hash table).
#include <limits.h>
#include <stdlib.h>

#include <type_traits>

#include "rubiks_common.h"

namespace rubiks {

constexpr int kA = kActions;   // 12
// A node's per-action rows live in ONE 256-byte record (rc_mcts_t, include/rubiks_hip.h), two cache lines:
//   line 0: N[12] | W[12] | walk record (16 B) | 16 spare   -- what a backup and a re-validation WRITE
//   line 1: P[12] | nbr[12] | 32 spare                      -- read-only once the node's children exist
// The struct's N / W / rec / P / nbr pointers point at their field of node 0, so entry a of node n is X[n * kRow + a] (rec as
// uint4: n * kRowRec).  Re-deciding a level of the previous path reads two adjacent lines and dirties one.
// The reference's L array (virtual losses, agents.py:427) is not stored at all: after every backup it is identically zero
// (the backup clears exactly the entries the descent raised, agents.py:569-570), so between iterations it IS the losses of
// the pending descent path -- path_node / path_act -- from which the descents here take their counts and from which
// MCTSForest.tree_arrays() reports it.
constexpr int kRow = RC_MCTS_NODE_WORDS, kRowRec = kRow / 4;
constexpr int kMaxPath = 4096;  // levels of a PUCT descent the select kernel stages in LDS (28 KiB per workgroup); deeper levels stay in HBM
// Level k of tree t in the blocked path arrays (path_node / path_act / path_next / short_act, see rc_mcts_t): block 0 is the dense
// [B][block] array, so for a shallow tree this is t * block + k.
__device__ __forceinline__ size_t path_at(const rc_mcts_t &m, u32 t, int k) {
    const u32 lg = m.path_block_log2;
    return ((((size_t)((u32)k >> lg)) * m.n_trees + t) << lg) + ((u32)k & ((1u << lg) - 1u));
}
// Levels of tree t's path arrays that have memory behind them.
__device__ __forceinline__ int path_limit(const rc_mcts_t &m, u32 t) {
    return m.path_rows ? min(m.path_rows[t], (int)m.max_path) : (int)m.max_path;
}

// Walk record of a node (16 B, rc_mcts_t::rec): what a PUCT descent does at the node while no virtual loss
// other than its own arrival edge is pending there.
//   x = neighbour through best0, y = neighbour through best1, z = best0 | best1 << 8 | kRecLeaf, w = line tag
//   best0 = argmax_a U(a) + W(a)                          (every L = 0)
//   best1 = the same with one virtual loss on best0       (the descent arrived through rev(best0))
// A leaf's record is {0, 0, kRecLeaf, 0}.  rc_mcts_select keeps the records exact (see k_mcts_select).
// Line tag: where the node last lay on a descent path -- path number (16 bits, 0 = never) << 16 | level << 4 |
// action taken there (15 = it was that path's leaf).  The last ring_k paths of a tree are kept in a ring; the tag
// only ever selects CANDIDATES for parallel validation, so a stale or aliased tag costs time, never correctness.
constexpr u32 kRecLeaf = 1u << 16;
constexpr u32 kNoAct = 15;
// (Round 3 built a "one-line" re-validation -- P[best0], P[best1] and the largest other P cached in the spare 16 bytes of line 0,
// 76 % of the levels re-decided without reading line 1, HBM fetch per launch 105 -> 64 MB -- and measured it SLOWER, 81.9 -> 97.8 us:
// the pass is bound by its instruction stream, not by HBM.  The code is gone; the measurement is profiles/r3_select_one_line_ab.txt.)

// Life of a tree slot.  A planted root is evaluated and expanded inside the ordinary lock-step iterations, on the 11 network
// rows every tree owns, in two steps (the root's expansion creates 12 new children, one more than the rows of a step):
//   ROOT_A  expand the root (children = nodes 2..13); rows: the root itself, then children 0..9
//   ROOT_B  rows: children 10, 11; then the root's backup (agents.py:555-561) and the first descent
// A solved child found by the root's expansion is only reported (status) when ROOT_B has completed the tree.
constexpr int kPhaseNormal = 0, kPhaseRootA = 1, kPhaseRootB = 2, kPhaseMask = 15, kPhaseSolved = 16;
__device__ __forceinline__ u32 line_tag(u32 seq, int level, u32 act) { return (seq << 16) | ((u32)level << 4) | act; }
// Tree served by workgroup `slot` (rc_mcts_t::active): -1 = nobody.  The tree's network rows are 11 slot .. 11 slot + 10.
__device__ __forceinline__ int tree_of(const rc_mcts_t &m, u32 slot) { return m.active ? m.active[slot] : (int)slot; }

// ---- plant: (re)start trees.  One workgroup per listed slot: hash table cleared, root = node 1, phase ROOT_A ------
__device__ __forceinline__ void expand_leaf_wave(const rc_mcts_t &m, u32 t, u32 slot, u32 lane, const u8 *lut, u32 max_states, bool root,
                                                 int leaf);

// expand_states > 0: the root's expansion (phase ROOT_A, normally the next rc_mcts_expand) happens here as well, with
// max_states = expand_states -- for the fused iteration rc_mcts_step*, which expands at the END of a step.  The tree's network
// rows are those of list position == tree index (the caller plants while every tree is listed in order).
__global__ __launch_bounds__(kBlock) void k_mcts_plant(rc_mcts_t m, const int *__restrict__ slots, const u8 *__restrict__ roots,
                                                     size_t stride, size_t first_col, u32 expand_states) {
    __shared__ u32 s_lut[sizeof(kTables.lut) / 4];
    __shared__ int s_solved;
    stage_to_lds(s_lut, c_tables.lut, sizeof(kTables.lut));
    const u32 t = slots ? (u32)slots[blockIdx.x] : blockIdx.x;
    const size_t col = first_col + blockIdx.x;
    uint4 *tab4 = reinterpret_cast<uint4 *>(m.hash + (size_t)t * m.hash_size);
    for (u32 i = threadIdx.x; i < m.hash_size / 4; i += kBlock) tab4[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (threadIdx.x == 0) {
    u32 w[4] = {0, 0, 0, 0};
    bool solved = true;
#pragma unroll
    for (int j = 0; j < kPlanes; ++j) {
        const u32 code = roots[(size_t)j * stride + col] & 31u;
        key_set(w, j, code);
        solved &= code == (u32)(u8)kTables.solved[j];
    }
    const uint4 key = make_uint4(w[0], w[1], w[2], w[3]);
    const size_t base = (size_t)t * (m.capacity + 1);
    reinterpret_cast<uint4 *>(m.keys)[base + 1] = key;
    m.hash[(size_t)t * m.hash_size + (key_hash(key) & (m.hash_size - 1))] = 1;
    m.leaf[base + 1] = 1;
    m.leaf[base] = 1;   // row 0 is never expanded; the reference's leaves[0] stays True as well
    reinterpret_cast<uint4 *>(m.rec)[base * kRowRec] = make_uint4(0, 0, kRecLeaf, 0);
    reinterpret_cast<uint4 *>(m.rec)[(base + 1) * kRowRec] = make_uint4(0, 0, kRecLeaf, 0);
    for (int a = 0; a < kA; ++a) {   // rows of the sentinel and of the root (children get theirs when they are created)
        m.N[base * kRow + a] = m.N[(base + 1) * kRow + a] = 0;
        m.nbr[base * kRow + a] = m.nbr[(base + 1) * kRow + a] = 0;
    }
    m.n_nodes[t] = 1;
    m.status[t] = solved ? RC_MCTS_ROOT_SOLVED : RC_MCTS_RUNNING;
    m.solved_idx[t] = solved ? 1 : -1;
    m.solved_action[t] = -1;
    m.iterations[t] = 0;
    m.path_len[t] = 1;
    m.pending[t] = 0;
    m.path_node[path_at(m, t, 0)] = 1;
    m.expanded[t] = 0;
    m.new_mask[t] = 0;
    m.phase[t] = kPhaseRootA;
    for (u32 i = 0; i < m.ring_k; ++i) m.ring_len[(size_t)t * m.ring_k + i] = 0;   // the previous tenant's lines are void
    s_solved = solved;
    }
    if (expand_states == 0) return;
    __threadfence();     // the root's key, its hash entry and the tree words are in memory before the wave below reads them
    __syncthreads();
    if (threadIdx.x < kWave && !s_solved)
        expand_leaf_wave(m, t, t, threadIdx.x, reinterpret_cast<const u8 *>(s_lut), expand_states, true, 1);
}

// ---- expand: one wave per tree, lane k < 12 owns child k ----------------------------------------
// The expansion of `leaf` (agents.py:505-544) by one full wave; the caller has established that the tree is running, not
// suspended and not in phase ROOT_B.  `lut`: the move table staged in LDS.
__device__ __forceinline__ void expand_leaf_wave(const rc_mcts_t &m, u32 t, u32 slot, u32 lane, const u8 *lut, u32 max_states, bool root,
                                                 int leaf) {
    const size_t base = (size_t)t * (m.capacity + 1);
    uint4 *keys = reinterpret_cast<uint4 *>(m.keys) + base;
    const size_t col0 = (size_t)m.rows_per_tree * slot;
    const int n = m.n_nodes[t];
    if ((u32)n + kA > max_states || (u32)n + kA > m.capacity) {   // agents.py:476
        if (lane == 0) m.status[t] = RC_MCTS_EXHAUSTED;
        return;
    }
    // node arrays mapped on demand: rows n + 1 .. n + 12 must have memory behind them.  If the host has not got that far the tree
    // sits this iteration out (nothing is written, `expanded` stays 0) and the descent of the next one ends at the same leaf.
    if (m.mapped_rows && n + kA >= m.mapped_rows[t]) return;
    int *tab = m.hash + (size_t)t * m.hash_size;
    const u32 mask = m.hash_size - 1;
    const uint4 pk = keys[leaf];
    const bool act = lane < kA;
    const u32 a = lane & (kActionPad - 1);

    // child k = action k on the leaf (agents.py:512-513)
    u32 w[4] = {0, 0, 0, 0};
    bool solved = act;
#pragma unroll
    for (int j = 0; j < kPlanes; ++j) {
        const u32 code = lut[a * (2 * kCodePad) + (j >= kCorners ? kCodePad : 0) + key_code(pk, j)];
        key_set(w, j, code);
        solved &= code == (u32)(u8)kTables.solved[j];
    }
    const uint4 ck = make_uint4(w[0], w[1], w[2], w[3]);

    // membership in this tree (agents.py:517-520): linear probing, full-key compare
    u32 h = key_hash(ck) & mask;
    int found = 0;
    if (act) {
        for (;;) {
            const int s = tab[h];
            if (s == 0) break;
            if (key_eq(keys[s], ck)) { found = s; break; }
            h = (h + 1) & mask;
        }
    }
    // unseen children take the next indices in child order (agents.py:523)
    const u64 newm = __ballot(act && found == 0);
    const int rank = __popcll(newm & ((1ull << lane) - 1ull));
    const int idx = found ? found : n + 1 + rank;
    // network input: the new children, packed in child order at columns 11 slot + rank (a non-root leaf has at most 11: its
    // parent is known); the root's step evaluates the root itself first, then children 0..9
    if (act && !found && rank < (root ? 10 : 11)) {
        const size_t col = col0 + (u32)rank + (root ? 1u : 0u);
#pragma unroll
        for (int j = 0; j < kPlanes; ++j) m.child_soa[(size_t)j * m.child_stride + col] = (int8_t)key_code(ck, j);
    }
    if (root && lane == kA) {
#pragma unroll
        for (int j = 0; j < kPlanes; ++j) m.child_soa[(size_t)j * m.child_stride + col0] = (int8_t)key_code(pk, j);
    }
    if (act && !found) {
        keys[idx] = ck;
        for (;;) {   // claim the first free slot from where the lookup stopped (siblings race here)
            if (atomicCAS(&tab[h], 0, idx) == 0) break;
            h = (h + 1) & mask;
        }
        m.leaf[base + idx] = 1;
        reinterpret_cast<uint4 *>(m.rec)[(base + idx) * kRowRec] = make_uint4(0, 0, kRecLeaf, 0);
        // a node's rows start here (no array is ever cleared wholesale): N = 0, neighbors = 0 but the way back
        uint4 *nrow = reinterpret_cast<uint4 *>(m.N + (base + idx) * kRow), *brow = reinterpret_cast<uint4 *>(m.nbr + (base + idx) * kRow);
        u32 back[kA];
#pragma unroll
        for (int e = 0; e < kA; ++e) back[e] = (u32)e == (lane ^ 1u) ? (u32)leaf : 0u;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            nrow[e] = make_uint4(0, 0, 0, 0);
            brow[e] = make_uint4(back[4 * e], back[4 * e + 1], back[4 * e + 2], back[4 * e + 3]);
        }
    }
    if (act) {   // links both ways, for seen children too (agents.py:533-535)
        m.nbr[(base + leaf) * kRow + lane] = idx;
        if (found) m.nbr[(base + idx) * kRow + (lane ^ 1)] = leaf;
        m.child_idx[(size_t)t * kA + lane] = idx;
    }
    const u64 solm = __ballot(solved);
    if (lane == 0) {
        m.leaf[base + leaf] = 0;
        m.n_nodes[t] = n + __popcll(newm);
        m.new_mask[t] = (u32)newm;
        m.expanded[t] = 1;
        m.iterations[t] += 1;
    }
    if (solm) {   // first solved child wins (agents.py:540-543)
        const int first = __ffsll((unsigned long long)solm) - 1;
        if ((int)lane == first) {
            if (root) m.phase[t] = kPhaseRootA | kPhaseSolved;   // reported once ROOT_B has completed the tree
            else m.status[t] = RC_MCTS_SOLVED;
            m.solved_idx[t] = idx;
            m.solved_action[t] = first;
        }
    }
}

// Second half of a root's iteration (phase ROOT_B): children 10 and 11 (nodes 12, 13) are this step's network rows.
__device__ __forceinline__ void expand_root_b(const rc_mcts_t &m, u32 t, u32 slot, u32 lane) {
    const uint4 *keys = reinterpret_cast<const uint4 *>(m.keys) + (size_t)t * (m.capacity + 1);
    const size_t col0 = (size_t)m.rows_per_tree * slot;
    if (lane < 2) {
        const uint4 ck = keys[2 + 10 + lane];
#pragma unroll
        for (int j = 0; j < kPlanes; ++j) m.child_soa[(size_t)j * m.child_stride + col0 + lane] = (int8_t)key_code(ck, j);
    }
    if (lane == 0) m.expanded[t] = 1;
}

__global__ __launch_bounds__(kWave) void k_mcts_expand(rc_mcts_t m, u32 max_states) {
    __shared__ u32 s_lut[sizeof(kTables.lut) / 4];
    stage_to_lds(s_lut, c_tables.lut, sizeof(kTables.lut));
    __syncthreads();
    const u8 *lut = reinterpret_cast<const u8 *>(s_lut);
    const u32 slot = blockIdx.x, lane = threadIdx.x;
    const int ti = tree_of(m, slot);
    if (ti < 0) return;
    const u32 t = (u32)ti;
    if (lane == 0) m.expanded[t] = 0;
    if (m.status[t] != RC_MCTS_RUNNING || m.pending[t]) return;   // a suspended descent has no leaf yet
    const int ph = m.phase[t] & kPhaseMask;
    if (ph == kPhaseRootB) {
        expand_root_b(m, t, slot, lane);
        return;
    }
    const int plen = m.path_len[t];
    expand_leaf_wave(m, t, slot, lane, lut, max_states, ph == kPhaseRootA, m.path_node[path_at(m, t, plen - 1)]);
}

// ---- backup: P/V of the new children, W/N/L along the path (agents.py:555-571) ------------------
__device__ __forceinline__ float wave_max12(float v, bool valid) {   // lanes 0..15 only matter
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(row16_max(valid ? v : -INFINITY))));
}

__device__ __forceinline__ float head_elem(const void *head, size_t i, bool bf16) {
    return bf16 ? __uint_as_float((u32) reinterpret_cast<const u16 *>(head)[i] << 16) : reinterpret_cast<const float *>(head)[i];
}

// HEAD = false: probs / values are separate float arrays (generic path, softmax done by the caller).
// HEAD = true : `probs` is the head GEMM's output (12 logits + value per row, bf16 or float) and the softmax
//               happens here.
// The children's part of the backup, by lanes 0..11 of one wave: P, V, W of the new children, W[leaf] = V[neighbors]
// (agents.py:555-561).  Returns the value that is backed up along the path (every lane).
template <bool HEAD>
__device__ __forceinline__ float net_row(const void *probs_or_head, const float *values, size_t row, size_t ld, bool head_bf16,
                                         float (&p)[kA]) {
    if (HEAD) {
        float mx = -INFINITY, sum = 0.f;
#pragma unroll
        for (int a = 0; a < kA; ++a) { p[a] = head_elem(probs_or_head, row * ld + a, head_bf16); mx = fmaxf(mx, p[a]); }
#pragma unroll
        for (int a = 0; a < kA; ++a) { p[a] = expf(p[a] - mx); sum += p[a]; }
#pragma unroll
        for (int a = 0; a < kA; ++a) p[a] /= sum;   // agents.py:552 softmax(dim=1)
        return head_elem(probs_or_head, row * ld + kA, head_bf16);
    }
    const float *probs = reinterpret_cast<const float *>(probs_or_head);
#pragma unroll
    for (int a = 0; a < kA; ++a) p[a] = probs[row * kA + a];
    return values[row];
}

template <bool HEAD>
__device__ __forceinline__ float backup_children(const rc_mcts_t &m, u32 t, u32 slot, u32 lane, size_t base, int leaf, int ph,
                                                 const void *probs_or_head, const float *values, size_t ld, bool head_bf16) {
    const bool act = lane < kA;
    const u32 newm = m.new_mask[t];
    const int idx = act ? m.child_idx[(size_t)t * kA + lane] : 0;
    const size_t row0 = (size_t)slot * m.rows_per_tree;
    // which children this step's rows hold: the new ones in child order -- or, for a root (all 12 new), 0..9 behind the
    // root's own row in ROOT_A and 10, 11 in ROOT_B
    const bool in_rows = ph == kPhaseRootA ? lane < 10 : ph == kPhaseRootB ? (act && lane >= 10) : (act && ((newm >> lane) & 1u));
    const size_t row = row0 + (ph == kPhaseRootA ? lane + 1 : ph == kPhaseRootB ? lane - 10 : (u32)__popc(newm & ((1u << lane) - 1u)));
    float v = 0.f;
    if (in_rows || (ph == kPhaseRootA && lane == kA)) {
        float p[kA];
        const bool is_root = !in_rows;
        v = net_row<HEAD>(probs_or_head, values, is_root ? row0 : row, ld, head_bf16, p);
        const size_t node = base + (is_root ? 1 : idx);
        m.V[node] = v;
#pragma unroll
        for (int a = 0; a < kA; ++a) {
            m.P[node * kRow + a] = p[a];
            if (!is_root) m.W[node * kRow + a] = v;   // W[new] = v for all 12 actions (agents.py:561)
        }
    } else if (act) {
        v = m.V[base + idx];
    }
    if (ph == kPhaseRootA) return 0.f;   // the root's own backup follows in ROOT_B, when all its children have values
    // best value among the new children (agents.py:559); with no new child the reference raises --
    // defined here as the best existing neighbour value (oracle/agents.py docstring)
    const float best = newm ? wave_max12(v, act && ((newm >> lane) & 1u)) : wave_max12(v, act);
    if (act) m.W[(base + leaf) * kRow + lane] = v;   // W[leaf] = V[neighbors[leaf]] (agents.py:560)
    return best;
}

// Phase bookkeeping after the children's part; returns true if the tree is done with this iteration (no path, no descent).
__device__ __forceinline__ bool backup_phase_done(const rc_mcts_t &m, u32 t, u32 tid, int phase) {
    const int ph = phase & kPhaseMask;
    if (ph == kPhaseRootA) {
        if (tid == 0) m.phase[t] = kPhaseRootB | (phase & kPhaseSolved);
        return true;
    }
    if (ph == kPhaseRootB) {
        if (tid == 0) {
            m.phase[t] = kPhaseNormal;
            if (phase & kPhaseSolved) m.status[t] = RC_MCTS_SOLVED;
        }
        return (phase & kPhaseSolved) != 0;
    }
    return false;
}

// Path updates of the backup (agents.py:562-570) by a whole workgroup.  NumPy's buffered `N[rows, cols] += 1` counts a
// (node, action) pair that occurs twice on the path only once; a mark bit reproduces that for any path length: first
// every path edge is marked, then whoever finds the mark replaces it by old + 1.  Two threads that hold the same pair
// write the same values, whichever of them runs first, so no ordering is needed inside a pass.
__device__ __forceinline__ void backup_path(const rc_mcts_t &m, u32 t, u32 tid, size_t base, int plen, float best, int nt = kBlock) {
    const int edges = plen - 1;
    constexpr int kMark = 1 << 30;
    for (int i = tid; i < edges; i += nt) {
        const size_t pi = path_at(m, t, i);
        m.N[(base + m.path_node[pi]) * kRow + m.path_act[pi]] |= kMark;
    }
    __syncthreads();
    for (int i = tid; i < edges; i += nt) {
        const size_t pi = path_at(m, t, i);
        const size_t e = (base + m.path_node[pi]) * kRow + m.path_act[pi];
        const int nv = m.N[e];
        if (nv & kMark) m.N[e] = (nv & ~kMark) + 1;            // agents.py:568
        m.W[e] = fmaxf(m.W[e], best);                           // agents.py:562
    }
}

template <bool HEAD>
__global__ __launch_bounds__(kBlock) void k_mcts_backup(rc_mcts_t m, const void *__restrict__ probs_or_head,
                                                      const float *__restrict__ values, size_t ld, bool head_bf16) {
    __shared__ float s_best;
    const u32 slot = blockIdx.x, tid = threadIdx.x;
    const int ti = tree_of(m, slot);
    if (ti < 0) return;
    const u32 t = (u32)ti;
    if (!m.expanded[t]) return;
    const size_t base = (size_t)t * (m.capacity + 1);
    const int plen = m.path_len[t];
    const int phase = m.phase[t];
    if (tid < kWave) {
        const float best = backup_children<HEAD>(m, t, slot, tid, base, m.path_node[path_at(m, t, plen - 1)], phase & kPhaseMask, probs_or_head,
                                                 values, ld, head_bf16);
        if (tid == 0) s_best = best;
    }
    __syncthreads();
    if (backup_phase_done(m, t, tid, phase)) return;
    backup_path(m, t, tid, base, plen, s_best);
}

// ---- select: PUCT descent with virtual loss (agents.py:575-595) ---------------------------------
// Scores of one node, NumPy's evaluation order in float64: ((c * P) * sqrt(sum N)) / (1 + N) + (W - L).
// Runs on a 16-lane DPP row (lanes 12..15 carry neutral elements); every lane of the row gets the result.
__device__ __forceinline__ int puct_argmax_sum(double c, int sum_n, int n_a, float p_f, float w_f, u32 l_cnt, bool act,
                                               int lane_in_row) {
    const double u = ((c * (double)p_f) * sqrt((double)sum_n)) / (double)(1 + n_a);
    const double score = act ? u + ((double)w_f - 100.0 * (double)l_cnt) : -INFINITY;
    return row16_argmax_first(score, act ? lane_in_row : 64);   // first maximum wins (agents.py:588)
}
__device__ __forceinline__ int puct_argmax(double c, int n_a, float p_f, float w_f, u32 l_cnt, bool act, int lane_in_row) {
    return puct_argmax_sum(c, row16_sum(act ? n_a : 0), n_a, p_f, w_f, l_cnt, act, lane_in_row);
}
// The sequential walk of k_mcts_select is one wave executing a dependent chain, so its speed is its
// instruction count (~4.7 cycles each).  The float64 argmax above is ~150 instructions; the walk evaluates
// the same formula in float32 (hardware sqrt / reciprocal, max by four DPP steps, winner by ballot) and
// ACCEPTS the float32 winner only if it leads every other action by more than the worst-case difference
// between the float32 and float64 evaluations (kPuctEps * the magnitudes involved: c, sqrt, reciprocal and
// four roundings are each within 2^-24 relative, 2^-20 leaves a factor of two).  Otherwise -- near ties
// and exact ties, where the first-maximum rule decides -- it falls back to the float64 evaluation.  The
// result is always the float64 argmax.  Lanes 0..11 of the wave hold the actions.
constexpr float kPuctEps = 0x1p-20f;
__device__ __forceinline__ int puct_argmax_walk(double c, float c32, int n_a, float p_f, float w_f, u32 l_cnt, bool act,
                                                int lane, int &slow) {
    const int sum_n = row16_sum_fused(act ? n_a : 0);
    const float loss = 100.0f * (float)l_cnt;   // exact: counts stay far below 2^24 / 100
    const float u = c32 * p_f * __builtin_amdgcn_sqrtf((float)sum_n) * __builtin_amdgcn_rcpf((float)(1 + n_a));
    const float mag = fabsf(u) + fabsf(w_f) + loss;
    const float score = act ? u + (w_f - loss) : -INFINITY;
    const float best = row16_max_fused(score);
    const u32 winners = (u32)__builtin_amdgcn_ballot_w64(score == best) & 0xFFFu;
    const int a = winners ? __builtin_ctz(winners) : 0;   // no winner: NaN scores, decided below in float64
    const float mag_a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mag), a));
    const bool close = act && lane != a && !(best - score > kPuctEps * (mag + mag_a));
    if ((((u32)__builtin_amdgcn_ballot_w64(close) & 0xFFFu) | (winners == 0)) == 0) return a;
    ++slow;
    return __builtin_amdgcn_readfirstlane(puct_argmax_sum(c, sum_n, n_a, p_f, w_f, l_cnt, act, lane));
}

// ---- PUCT of one node evaluated by ONE lane (all twelve actions in registers), float32 ------------------------
// Same arithmetic and the same acceptance rule as puct_argmax_walk: the float32 winner is taken only if it leads
// every other action by more than the worst-case float32-vs-float64 difference; ties, near ties and NaNs report
// `certain = false` and are decided in float64 by the caller.  One lane per tree level turns the re-validation of
// a 1 000-level path into four passes of a 256-thread workgroup.
struct LaneEval {
    float u[kA], w[kA];
};
__device__ __forceinline__ void load_row12(const void *base, size_t row, u32 (&out)[kA]) {
    const uint4 *p = reinterpret_cast<const uint4 *>(reinterpret_cast<const u32 *>(base) + row);   // rows are 48 B, 16-B aligned
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const uint4 v = p[i];
        out[4 * i] = v.x, out[4 * i + 1] = v.y, out[4 * i + 2] = v.z, out[4 * i + 3] = v.w;
    }
}
__device__ __forceinline__ void store_row12(void *base, size_t row, const u32 (&v)[kA]) {
    uint4 *p = reinterpret_cast<uint4 *>(reinterpret_cast<u32 *>(base) + row);
#pragma unroll
    for (int i = 0; i < 3; ++i) p[i] = make_uint4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
}
__device__ __forceinline__ LaneEval lane_eval(float c32, const u32 (&n)[kA], const u32 (&p)[kA], const u32 (&w)[kA]) {
    int sum = 0;
#pragma unroll
    for (int a = 0; a < kA; ++a) sum += (int)n[a];
    const float csq = c32 * __builtin_amdgcn_sqrtf((float)sum);   // c sqrt(sum N), once per node
    LaneEval e;
#pragma unroll
    for (int a = 0; a < kA; ++a) {
        e.u[a] = __uint_as_float(p[a]) * csq * __builtin_amdgcn_rcpf((float)(1 + (int)n[a]));
        e.w[a] = __uint_as_float(w[a]);
    }
    return e;
}
__device__ __forceinline__ LaneEval lane_prepare(float c32, const rc_mcts_t &m, size_t row) {
    u32 n[kA], p[kA], w[kA];
    load_row12(m.N, row, n);
    load_row12(m.P, row, p);
    load_row12(m.W, row, w);
    return lane_eval(c32, n, p, w);
}
// cnt5: virtual-loss count of action a in bits 5a .. 5a+4
__device__ __forceinline__ int lane_pick(const LaneEval &e, u64 cnt5, bool &certain) {
    float sc[kA], mg[kA];
    float best = -INFINITY, mbest = 0.f;
    int arg = 0;
    bool ok = true;
#pragma unroll
    for (int a = 0; a < kA; ++a) {
        const float loss = 100.0f * (float)((u32)(cnt5 >> (5 * a)) & 31u);
        sc[a] = e.u[a] + (e.w[a] - loss);
        mg[a] = fabsf(e.u[a]) + fabsf(e.w[a]) + loss;
        ok &= sc[a] == sc[a];
        if (sc[a] > best) { best = sc[a]; arg = a; mbest = mg[a]; }
    }
#pragma unroll
    for (int a = 0; a < kA; ++a) ok &= (a == arg) | (best - sc[a] > kPuctEps * (mg[a] + mbest));
    certain = ok;
    return arg;
}
// The two picks every re-decided level needs -- best0 (no loss) and best1 (one loss on best0) -- from ONE pass over the twelve
// scores: the three largest scores (a min / max cascade), the actions of the two largest, the largest magnitude and the sum of
// the scores (NaN detector).  best1 differs from best0 in one entry, so it is either best0 itself (its score less 100 still
// leads) or the runner-up.  Acceptance is the rule of lane_pick made one-sided: the winner must lead the next score by more than
// kPuctEps x twice the LARGEST magnitude of the node (>= the two magnitudes lane_pick adds), so whatever is accepted here
// lane_pick accepts too, with the same winner; ties, near ties, NaNs and infinities of opposite sign come back uncertain and
// are re-decided in float64.  (score of best0 with its loss = best - 100: one rounding more than u + (w - 100), far inside the
// margin, which is then taken on magnitudes + 100.)  ~190 VALU instructions per level less than two full argmax passes.
__device__ __forceinline__ void lane_pick2(const LaneEval &e, int &b0, bool &c0, int &b1, bool &c1) {
    float best = -INFINITY, second = -INFINITY, third = -INFINITY, mmax = 0.f, acc = 0.f;
    int arg = 0, arg2 = 0;
#pragma unroll
    for (int a = 0; a < kA; ++a) {
        const float sc = e.u[a] + e.w[a];
        mmax = fmaxf(mmax, fabsf(e.u[a]) + fabsf(e.w[a]));
        acc += sc;
        const bool lead = sc > best, runner = sc > second;     // first maximum keeps its place on ties (which are uncertain anyway)
        arg2 = lead ? arg : runner ? a : arg2;
        arg = lead ? a : arg;
        const float t1 = fminf(sc, best);
        best = fmaxf(best, sc);
        const float t2 = fminf(t1, second);
        second = fmaxf(second, t1);
        third = fmaxf(third, t2);
    }
    const bool finite = acc == acc;                              // a NaN score (or +inf and -inf) poisons the sum
    b0 = arg;
    c0 = finite & (best - second > kPuctEps * 2.f * mmax);
    const float lost = best - 100.0f;                            // best0 carrying one virtual loss
    const bool stays = lost > second;
    b1 = stays ? arg : arg2;
    const float lead1 = stays ? lost - second : second - fmaxf(third, lost);
    c1 = finite & (lead1 > kPuctEps * 2.f * (mmax + 100.0f));
}
__device__ __forceinline__ void cnt5_add(u64 &cnt5, bool &overflow, u32 a) {
    overflow |= ((u32)(cnt5 >> (5 * a)) & 31u) >= 30u;
    cnt5 += 1ull << (5 * a);
}

// Per-tree arrays as buffer resources (wave-uniform, in SGPRs): a row access is one instruction with one
// shared 32-bit offset register instead of 64-bit address arithmetic per array.
constexpr u32 kRsrcFlags = 0x00020000;   // raw buffer, 32-bit data format
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
struct TreeBufs {
    __amdgpu_buffer_rsrc_t N, nbr, P, W, rec;
};
__device__ __forceinline__ TreeBufs tree_bufs(const rc_mcts_t &m, size_t base) {
    const u32 rows = (m.capacity + 1) * (u32)kRow;   // words; x 4 bytes must stay below 2^32: capacity + 1 < 2^24 (check_mcts)
    TreeBufs b;
    b.N = __builtin_amdgcn_make_buffer_rsrc((void *)(m.N + base * kRow), 0, (int)(rows * 4u), kRsrcFlags);
    b.nbr = __builtin_amdgcn_make_buffer_rsrc((void *)(m.nbr + base * kRow), 0, (int)(rows * 4u), kRsrcFlags);
    b.P = __builtin_amdgcn_make_buffer_rsrc((void *)(m.P + base * kRow), 0, (int)(rows * 4u), kRsrcFlags);
    b.W = __builtin_amdgcn_make_buffer_rsrc((void *)(m.W + base * kRow), 0, (int)(rows * 4u), kRsrcFlags);
    b.rec = __builtin_amdgcn_make_buffer_rsrc((void *)(reinterpret_cast<uint4 *>(m.rec) + base * kRowRec), 0, (int)(rows * 4u),
                                              kRsrcFlags);
    return b;
}
// The rows of one node as a full PUCT evaluation needs them (lane a < 12 holds action a's entries).
struct NodeRows {
    int n_a, nb;
    float p_f, w_f;
};
__device__ __forceinline__ NodeRows load_rows(const TreeBufs &tb, int node, u32 la) {
    const u32 off = ((u32)node * kRow + la) * 4u;
    NodeRows x;
    x.n_a = (int)__builtin_amdgcn_raw_buffer_load_b32(tb.N, off, 0, 0);
    x.p_f = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(tb.P, off, 0, 0));
    x.w_f = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(tb.W, off, 0, 0));
    x.nb = (int)__builtin_amdgcn_raw_buffer_load_b32(tb.nbr, off, 0, 0);
    return x;
}
__device__ __forceinline__ u32x4 load_rec(const TreeBufs &tb, int node) {
    return __builtin_amdgcn_raw_buffer_load_b128(tb.rec, (u32)node * (kRow * 4u), 0, 0);
}


constexpr int kSelHash = 2048;   // LDS chain heads of the select kernel
__device__ __forceinline__ u32 sel_hash(int node) { return ((u32)node * 0x9E3779B1u) >> 21; }

// One 256-thread workgroup per tree.
//
// The descent restarts at the root every iteration (that is the algorithm), and with a trained network it is
// hundreds of levels deep, so its cost per level is what bounds a tree's iteration rate.  Two facts make it cheap:
//
// (1) Between descents every virtual loss is zero (the backup clears the path's), and inside a descent the only
//     loss pending at a node the descent has NOT been at before is the one on its arrival edge.  So what the
//     descent does at such a node is a function of the node's rows alone: best0 (no loss) unless it arrived through
//     rev(best0), then best1 (one loss on best0).  Both, and the neighbours they lead to, are kept in a 16-byte
//     walk record per node.  A node's rows change only while it is on the path being backed up, so refreshing the
//     records of the previous path's nodes (below, in parallel) keeps every record of the tree exact.
// (2) Consecutive descents share long prefixes.  Level k of the previous path keeps its action iff its decision is
//     unchanged GIVEN that all levels above kept theirs -- a premise under which its pending losses are a function
//     of the old path alone.  So all levels are re-decided in parallel (one 16-lane row per level, the same pass
//     that refreshes the records); the first level whose action changes starts the sequential walk.
//
// A node the descent revisits (deep lines loop through transpositions all the time) carries more losses: every
// earlier visit j left +1 on its departure edge a_j and +1 on its arrival edge rev(a_{j-1}).  Those levels are
// decided by a full float64 PUCT evaluation with the exact counts, taken from the path itself, which is chained
// by node in an LDS hash table.  The sequential walk is therefore one 16-byte load per level (the next record,
// requested before the revisit test of the current level is done) and no stores; path and virtual losses go
// to memory once, after the loop.
//
// MODE 1 / 2 (rc_mcts_backup_select*): the kernel also IS the backup of the iteration (MODE 2: from the network's head
// output, softmax inside).  The children's part runs first on one wave; the path part needs no pass of its own: the lane
// that re-decides level k holds the node's N / P / W rows in registers anyway, applies the visit and the max-backup of
// EVERY level of the path at that node to them (the chain of the node lists those levels; a repeated (node, action) pair
// counts once, as NumPy's buffered += does), evaluates the updated rows and -- if it is the node's first level -- writes
// them back.  One kernel, one pass over the path and one memory round trip less per iteration.
// FUSE (rc_mcts_step*): the kernel is the whole tree side of an iteration -- backup of the previous expansion, descent, and the
// EXPANSION of the leaf the descent ends at (what rc_mcts_expand would do at the start of the next iteration), by the wave that
// walked there: one launch and one dependent kernel boundary less per iteration.  Trees planted for this form have their root
// expanded by rc_mcts_plant_expanded.
// NT: threads per tree.  256 in a full forest (four workgroups per CU: 1 024 trees in one round); small forests leave CUs idle, so
// rc_mcts_step* gives a tree 512 / 1 024 threads there -- the parallel parts (staging, re-validation: one lane per path level)
// of a 1 200-level descent take two rounds per thread instead of five.
// LW: waves that check a line together (1: wave 0 alone, 64 levels per round; 4: waves 0-3, 256 levels per round -- see "line round").
template <int MODE, bool FUSE = false, int NT = kBlock, int LW = 1>
__global__ __launch_bounds__(NT) void k_mcts_select(rc_mcts_t m, double c, u32 level_budget, const void *__restrict__ probs_or_head,
                                                      const float *__restrict__ values, size_t ld, bool head_bf16, u32 max_states) {
    __shared__ u32 s_lut[FUSE ? sizeof(kTables.lut) / 4 : 1];
    __shared__ float s_best;                // MODE > 0: the value backed up along the path
    __shared__ int s_first;                 // first level that has to be walked sequentially
    __shared__ int s_node[kMaxPath];        // the path: old levels, then the walked ones
    __shared__ u8 s_act[kMaxPath];
    __shared__ int s_head[kSelHash];        // chains of path levels by node (earlier visits of a state)
    __shared__ u16 s_next[kMaxPath];
    // 2.5 KiB used twice.  Re-validation: two bitmaps (levels float32 could not settle; later levels of nodes the path visits more
    // than once) and the same two sets as dense lists (so that their passes use every lane), as long as they fit -- what does not
    // fit is found through the bitmaps.  Walk: the lane lists of a line segment (64 lanes, or 256 with LW = 4).
    __shared__ u32 s_pool[640];
    u32 *s_scratch = s_pool, *s_unc = s_pool + 384, *s_late = s_pool + 512;
    __shared__ int s_nlate, s_nunc;
    __shared__ int s_mb[64];                // LW > 1: mailbox between the walking wave and its helpers
    constexpr int kLateCap = 640, kUncCap = 128;
    const u32 unc_cap = min(m.unc_list_cap, (u32)kUncCap);   // tests lower it to reach the bitmap form of pass B with few flagged levels
    u16 *s_latelist = reinterpret_cast<u16 *>(s_scratch), *s_unclist = s_latelist + kLateCap;
    int *s_seg = reinterpret_cast<int *>(s_scratch);               // [256] last lane per node bucket of a candidate segment ...
    uint2 *s_segent = reinterpret_cast<uint2 *>(s_scratch + 256);   // [64] ... and per lane {node, previous lane | action << 8 | rev(arrival) << 12}
    const u32 slot = blockIdx.x, tid = threadIdx.x;
    const int ti = tree_of(m, slot);
    if (ti < 0) return;
    const u32 t = (u32)ti;
    const bool running = m.status[t] == RC_MCTS_RUNNING;
    const bool backup = MODE > 0 && m.expanded[t];   // uniform over the workgroup
    if (!running && !backup) return;
    if (FUSE) {
        stage_to_lds(s_lut, c_tables.lut, sizeof(kTables.lut));
        __syncthreads();                       // every thread has read `expanded`: it is consumed here (the expansion at the
        if (tid == 0) m.expanded[t] = 0;       // end of this call raises it again for the next one)
    }
    const unsigned long long t_begin = wall_clock64();
    const u32 seq = (u32)m.iterations[t] & 0xFFFFu;   // number of the path this call builds (its expansion count)
    const size_t base = (size_t)t * (m.capacity + 1);
    uint4 *rec = reinterpret_cast<uint4 *>(m.rec) + base * kRowRec;   // record of node n: rec[n * kRowRec]
    const int plen_old = m.path_len[t];
    const int phase = m.phase[t];
    // The path this call works on: levels below W in LDS (s_node / s_act, chained by node through s_head / s_next), deeper levels
    // where they lie in HBM (path_node / path_act, chained through path_next).  The reference's descent has no length limit
    // (agents.py:575-595); a tree that deep is rare, so its deep levels cost a memory round trip where the others cost an LDS read.
    // The body below exists twice: DEEP = false for a path that lies in LDS whole (every level access is the plain LDS access of
    // rounds 1-5: a test per access costs the sequential walk ~20 %, profiles/r6_select_deep_ab.txt) -- its walk stops at the
    // window's end and the descent is suspended there for one iteration --, DEEP = true for a path that is, or has become, deeper.
    auto body = [&](auto deep_c) __attribute__((always_inline)) {
    constexpr bool DEEP = decltype(deep_c)::value;
    const int W = DEEP ? (int)m.lds_levels : kMaxPath;
    auto P = [&](int k) { return path_at(m, t, k); };
    auto node_at = [&](int k) -> int { return (!DEEP || k < W) ? s_node[k] : m.path_node[P(k)]; };
    auto act_at = [&](int k) -> u32 { return (!DEEP || k < W) ? (u32)s_act[k] : (u32)m.path_act[P(k)]; };
    auto next_at = [&](int k) -> int {
        if (!DEEP || k < W) {
            const u32 nx = s_next[k];
            return nx == 0xFFFFu ? -1 : (int)nx;
        }
        return (int)m.path_next[P(k)];
    };
    auto put_node = [&](int k, int node) { if (!DEEP || k < W) s_node[k] = node; else m.path_node[P(k)] = node; };
    auto put_act = [&](int k, u32 a) { if (!DEEP || k < W) s_act[k] = (u8)a; else m.path_act[P(k)] = (u8)a; };
    auto put_next = [&](int k, int nx) { if (!DEEP || k < W) s_next[k] = (u16)nx; else m.path_next[P(k)] = (u32)nx; };   // (u16)-1 = 0xFFFF = none
    // All levels lo .. hi - 1 enter the chains of their nodes: first the LDS levels, then the deep ones, so that the 16-bit links
    // of the LDS levels only ever name LDS levels (or levels a line round appends right behind them).
    auto chain_levels = [&](int lo, int hi) {
        for (int k = lo + (int)tid; k < min(hi, W); k += NT) s_next[k] = (u16)atomicExch(&s_head[sel_hash(s_node[k])], k);
        if (DEEP && hi > W) {
            __syncthreads();
            for (int k = max(lo, W) + (int)tid; k < hi; k += NT) {
                const size_t pk = P(k);
                m.path_next[pk] = (u32)atomicExch(&s_head[sel_hash(m.path_node[pk])], k);
            }
        }
    };
    if (MODE == 0 && (phase & kPhaseMask) != kPhaseNormal) return;   // a root's first descent follows its backup in ROOT_B
    if (MODE > 0 && backup) {
        if (tid < kWave) {
            const float best = backup_children<MODE == 2>(m, t, slot, tid, base, m.path_node[P(plen_old - 1)], phase & kPhaseMask, probs_or_head,
                                                           values, ld, head_bf16);
            if (tid == 0) s_best = best;
        }
        if ((phase & kPhaseMask) != kPhaseNormal) {   // uniform over the workgroup
            __syncthreads();
            if (backup_phase_done(m, t, tid, phase)) {
                if (FUSE && (phase & kPhaseMask) == kPhaseRootA && tid < kWave) expand_root_b(m, t, slot, tid);   // the next step's rows
                return;
            }
        }
        if (!running) {   // the expansion ended the tree (a solved child): its backup is all that is left to do
            __syncthreads();
            backup_path(m, t, tid, base, plen_old, s_best, NT);
            return;
        }
    } else if (MODE > 0 && (phase & kPhaseMask) != kPhaseNormal) {
        return;
    }
    const int nlev = plen_old - 1;     // levels 0 .. nlev - 1 carry an action; level nlev is the old leaf
    // The re-validation below is bound by fetching the records of the old path's nodes from HBM (two cache lines per level, a
    // different node each), and a thread handles levels tid, tid + 256, ... one after the other: a deep tree would pay the
    // memory latency once per 256 levels while the shallow trees are long done.  So every thread requests the lines of ALL its
    // levels now (first 1 024 levels); the values are only consumed after the pass, the later rounds of the pass find their
    // records in L2.
    constexpr int kPf = 4;
    u32 pf[2 * kPf];
#pragma unroll
    for (int i = 0; i < kPf; ++i) {
        const int k = (int)tid + i * NT;
        const u32 *rp = reinterpret_cast<const u32 *>(m.N) + (base + (size_t)(k < plen_old ? m.path_node[P(k)] : 0)) * kRow;
        pf[2 * i] = rp[0];
        pf[2 * i + 1] = rp[kRow / 2];
    }
    const u32 row = tid >> 4, rl = tid & 15;
    const bool ract = rl < kA;
    const u32 rla = ract ? rl : 0;
#ifdef RUBIKS_SELECT_PHASES
    unsigned long long ph1 = 0, ph2 = 0, ph3 = 0;   // diagnostic build: ticks at the end of staging, pass A (+ late levels), pass B
#endif
    const int resume = m.pending[t];   // uniform over the workgroup: a suspended descent continues at its last node
    for (int i = tid; i < kSelHash; i += NT) s_head[i] = -1;
    for (int k = tid; k < min(plen_old, W); k += NT) {
        const size_t pk = P(k);
        s_node[k] = m.path_node[pk];
        s_act[k] = (k < nlev) ? m.path_act[pk] : (u8)0;
    }
    if (tid == 0) s_first = nlev;
    __syncthreads();
    chain_levels(0, nlev);
    __syncthreads();
#ifdef RUBIKS_SELECT_PHASES
    ph1 = wall_clock64();
#endif
    if (!resume) {
        // Pass A: one lane per level, float32 with the acceptance rule of lane_pick.  Levels it cannot settle (near
        // ties, NaNs, loss counts beyond 5 bits) are flagged for pass B.
        // The flags (bitmaps and lists in LDS) hold kMaxPath levels: a deeper path is re-validated kMaxPath levels at a time,
        // window after window (a level's decision depends on the levels ABOVE it only through the premise that they kept
        // their actions, and a node's rows are written by its FIRST level, which lies in this window or an earlier one).
        const float c32v = (float)c;
        const float best_up = MODE > 0 ? s_best : 0.f;
      for (int wb = 0; wb <= nlev; wb += kMaxPath) {
        const int wend = min(nlev, wb + kMaxPath - 1);   // last level of the window
        for (int i = tid; i < kMaxPath / 32; i += NT) s_unc[i] = s_late[i] = 0;
        if (tid == 0) s_nlate = s_nunc = 0;
        __syncthreads();
        // late = false: every level (MODE 0), or the FIRST level of every node (MODE > 0), which also applies the backup
        //               to the node's rows and writes them back;
        // late = true : (MODE > 0) the later levels of nodes the path visits more than once, after a barrier: they must
        //               read the rows as the first level left them.
        auto decide = [&](int k, bool late) {
            const int node = node_at(k);
            const int arr = k > 0 ? (int)(act_at(k - 1) ^ 1u) : -1;   // own arrival edge
            u64 cnt5 = 0;
            bool dup = false, ovf = false;
            u32 taken = 0;      // MODE > 0: actions the path takes at this node, over all its levels
            int first_lvl = k;  // ... and the first of those levels
            for (int j = s_head[sel_hash(node)]; j >= 0; j = next_at(j)) {
                if (node_at(j) == node) {
                    taken |= 1u << act_at(j);
                    first_lvl = min(first_lvl, j);
                    if (j < k) {
                        dup = true;
                        cnt5_add(cnt5, ovf, act_at(j));
                        if (j > 0) cnt5_add(cnt5, ovf, act_at(j - 1) ^ 1u);
                    }
                }
            }
            const int kw = k - wb;   // the level's place in the window's flags
            if (MODE > 0 && backup && !late && k != first_lvl) {
                const int pos = atomicAdd(&s_nlate, 1);
                if (pos < kLateCap) s_latelist[pos] = (u16)kw;
                else atomicOr(&s_late[kw >> 5], 1u << (kw & 31));
                return;
            }
            const size_t r = (base + node) * kRow;
            u32 n[kA], p[kA], w[kA];
            load_row12(m.N, r, n);
            load_row12(m.W, r, w);
            if (MODE > 0 && backup && !late && taken) {   // N[path, a] += 1 (once per pair), W[path, a] = max(W, best)  (agents.py:562,568)
#pragma unroll
                for (int a = 0; a < kA; ++a)
                    if ((taken >> a) & 1u) {
                        n[a] += 1u;
                        w[a] = __float_as_uint(fmaxf(__uint_as_float(w[a]), best_up));
                    }
                store_row12(m.N, r, n);
                store_row12(m.W, r, w);
            }
            if (dup && arr >= 0) cnt5_add(cnt5, ovf, (u32)arr);
            bool c0 = false, c1 = false, c2 = true;
            int b0 = 0, b1 = 0, d = 0;
            load_row12(m.P, r, p);
            const LaneEval e = lane_eval(c32v, n, p, w);
            lane_pick2(e, b0, c0, b1, c1);
            d = (arr == b0) ? b1 : b0;
            if (dup) d = lane_pick(e, cnt5, c2);
            if (c0 && c1 && c2 && !ovf) {
                const u32 nb0 = (u32)m.nbr[r + b0], nb1 = (u32)m.nbr[r + b1];
                const u32 a_k = k < nlev ? act_at(k) : kNoAct;
                // the old path is line seq - 1 (its first ring_levels levels: the tag has 12 bits of level; deeper levels keep the
                // tag they have, which then names some older line -- a tag only ever proposes candidates)
                const u32 tagw = k < (int)m.ring_levels ? line_tag((seq - 1) & 0xFFFFu, k, a_k) : reinterpret_cast<const u32 *>(rec + (size_t)node * kRowRec)[3];
                rec[(size_t)node * kRowRec] = make_uint4(nb0, nb1, (u32)b0 | ((u32)b1 << 8), tagw);
                if (k < nlev && d != (int)a_k) atomicMin(&s_first, k);
            } else {
                const int pos = atomicAdd(&s_nunc, 1);
                if (pos < kUncCap) s_unclist[pos] = (u16)kw;   // (the list is written up to its size; pass B reads it only while nunc <= unc_cap)
                atomicOr(&s_unc[kw >> 5], 1u << (kw & 31));
            }
        };
        for (int k = wb + (int)tid; k <= wend; k += NT) decide(k, false);
        if (MODE > 0 && backup) {
            __syncthreads();
#ifdef RUBIKS_SELECT_PHASES
            ph3 = wall_clock64();   // (overwritten by pass B's stamp unless RUBIKS_SELECT_PHASES == 2: first levels | late levels)
#endif
            const int nlate = s_nlate;
            for (int i = tid; i < min(nlate, kLateCap); i += NT) decide(wb + (int)s_latelist[i], true);
            if (nlate > kLateCap)   // the overflow, by the bitmap: each group of 32 lanes takes a word of flags
                for (int k0 = 32 * (int)(tid >> 5); wb + k0 <= wend; k0 += 32 * (NT / 32))
                    if ((s_late[k0 >> 5] >> (tid & 31)) & 1u) decide(wb + k0 + (int)(tid & 31), true);
        }
        __syncthreads();
#ifdef RUBIKS_SELECT_PHASES
        ph2 = wall_clock64();
#endif
        // Pass B: the flagged levels in float64, NumPy's evaluation order, one 16-lane row per level.
        const int nunc = s_nunc;
        const bool by_list = nunc <= (int)unc_cap;   // else: scan the bitmap, kStep = NT / 16 levels per step (one 16-lane row each)
        constexpr int kStep = NT / 16;                // 16 / 32 / 64: a quarter of a flag word, one word, two words
        for (int i0 = 0; i0 < (by_list ? nunc : wend - wb + 1); i0 += kStep) {
            if (!by_list) {                           // skip steps without a flagged level (uniform over the workgroup)
                const u32 w0 = s_unc[i0 >> 5];
                const u32 any = kStep == 16 ? ((w0 >> (i0 & 31)) & 0xFFFFu) : kStep == 32 ? w0 : (w0 | s_unc[(i0 >> 5) + 1]);
                if (any == 0) continue;
            }
            const int i = i0 + (int)row;
            const int kw = by_list ? (i < nunc ? (int)s_unclist[i] : kMaxPath) : i;   // place in the window (kMaxPath = none)
            const int k = wb + kw;
            const bool live = kw < kMaxPath && k <= wend && ((s_unc[kw >> 5] >> (kw & 31)) & 1u);
            const int node = live ? node_at(k) : 0;
            const u32 arr_k = (live && k > 0) ? (act_at(k - 1) ^ 1u) : 255u;
            const u32 act_k = (live && k < nlev) ? act_at(k) : kNoAct;
            u32 cnt = arr_k == rl ? 1u : 0u;   // own arrival edge
            bool dup = false;
            int j = live ? s_head[sel_hash(node)] : -1;
            while (j >= 0) {
                if (j < k && node_at(j) == node) {
                    dup = true;
                    cnt += act_at(j) == rl;
                    if (j > 0) cnt += (act_at(j - 1) ^ 1u) == rl;
                }
                j = next_at(j);
            }
            const size_t r = (base + node) * kRow + rla;
            const int n_a = m.N[r], nb = m.nbr[r];
            const float p_f = m.P[r], w_f = m.W[r];
            const int sum_n = row16_sum(ract ? n_a : 0);
            const double ud = ((c * (double)p_f) * sqrt((double)sum_n)) / (double)(1 + n_a);
            const double s0 = ract ? ud + ((double)w_f - 100.0 * 0.0) : -INFINITY;
            const int idx = ract ? (int)rl : 64;
            const int b0 = row16_argmax_first(s0, idx);
            const double s1 = ((int)rl == b0) ? ud + ((double)w_f - 100.0 * 1.0) : s0;
            const int b1 = row16_argmax_first(s1, idx);
            const double sx = ract ? ud + ((double)w_f - 100.0 * (double)cnt) : -INFINITY;
            const int bx = row16_argmax_first(sx, idx);
            const int d = dup ? bx : (int)arr_k == b0 ? b1 : b0;
            const u32 lane0 = (tid & 63u) & ~15u;
            const int nb0 = __builtin_amdgcn_ds_bpermute((int)((lane0 + (u32)b0) << 2), nb);
            const int nb1 = __builtin_amdgcn_ds_bpermute((int)((lane0 + (u32)b1) << 2), nb);
            if (live && rl == 0) {
                const u32 tagw = k < (int)m.ring_levels ? line_tag((seq - 1) & 0xFFFFu, k, act_k) : reinterpret_cast<const u32 *>(rec + (size_t)node * kRowRec)[3];
                rec[(size_t)node * kRowRec] = make_uint4((u32)nb0, (u32)nb1, (u32)b0 | ((u32)b1 << 8), tagw);
                if (k < nlev && d != (int)act_k) atomicMin(&s_first, k);
            }
        }
        __syncthreads();
#ifdef RUBIKS_SELECT_PHASES
        if (RUBIKS_SELECT_PHASES != 2) ph3 = wall_clock64();
#endif
      }
        u32 pfx = 0;
#pragma unroll
        for (int i = 0; i < 2 * kPf; ++i) pfx ^= pf[i];
        if (pfx == 0x5EED1234u && nlev < 0) s_first = 0;   // never true: keeps the requests above alive
        const int first = s_first;
        // (the kept prefix's virtual losses, agents.py:589-591, are implied by the path: see the note on L at the top)
        if (first < nlev) {   // from here on the chains hold the kept levels only; the walk appends its own
            for (int i = tid; i < kSelHash; i += NT) s_head[i] = -1;
            __syncthreads();
            chain_levels(0, first);
        }
    }
    __syncthreads();   // also: the records written above are visible to wave 0 from here on
    if (tid >= LW * kWave) return;

    // ---- line round with LW waves (LW > 1) ---------------------------------------------------------------------------
    // The walking wave (wave 0) publishes a line position in the mailbox; waves 0 .. LW - 1 then check 64 LW consecutive
    // levels of that line at once, lane gl = 64 wave + lane taking level kb + 1 + gl, exactly as the single-wave round below
    // does for 64 (same decisions, same acceptance rules): the levels' earlier visits are counted in the path's chains and,
    // for the lanes above in the round, in bucket lists all 64 LW lanes enter before anyone reads them (one barrier); where
    // the predecessor's line action leads is read from the predecessor's row by the lane itself, so no value crosses a wave
    // except through the mailbox: per wave the first failing lane, and what the walking wave needs to go on from there.
    // Barriers: the helpers wait at the top of their loop; every command of the walking wave is one barrier, a round three more.
    enum { MB_CMD = 0, MB_LINE_LO, MB_LINE_HI, MB_LLEN, MB_LPOS, MB_KB, MB_WANT, MB_ARR0, MB_ROOM, MB_Q = 12, MB_LAST_NL = 16, MB_LAST_ACT = 20,
           MB_STOP_NL = 24, MB_STOP_ACT = 28, MB_STOP_NODE = 32, MB_STOP_REC = 36 };
    uint2 *s_segent4 = reinterpret_cast<uint2 *>(s_pool);          // [256] per lane {node, previous lane | action << 16 | rev(arrival) << 20}
    int *s_seg4 = reinterpret_cast<int *>(s_pool + 512);           // [128] last lane per node bucket (0xFFFF = none)
    auto line_round = [&](const u32 wv, const u32 lane, const TreeBufs &tb, const float c32) -> int {   // returns the levels appended
        const size_t line = ((size_t)(u32)s_mb[MB_LINE_HI] << 32) | (u32)s_mb[MB_LINE_LO];
        const int llen = s_mb[MB_LLEN], lpos = s_mb[MB_LPOS], kb = s_mb[MB_KB], want = s_mb[MB_WANT], room = s_mb[MB_ROOM];
        const u32 arr0 = (u32)s_mb[MB_ARR0];
        const int gl = (int)(wv * kWave + lane);
        const bool in_line = lpos + gl < llen && gl < room;
        const int node_i = in_line ? m.ring_node[line + lpos + gl] : 0;
        const u32 act_i = in_line ? (u32)m.ring_act[line + lpos + gl] : kNoAct;
        const bool has_pred = in_line && gl > 0;
        const u32 arr_i = has_pred ? (u32)m.ring_act[line + lpos + gl - 1] : arr0;
        const u32x4 r = load_rec(tb, node_i);
        const bool inner = in_line && act_i != kNoAct;
        // Where the line's action leads is the line's next node (a neighbour entry never changes once it is set, and the line is a path
        // that was walked): read from the ring, not from the node's neighbour row -- a level then costs ONE line of its node's record
        // (the walk record) instead of two, and the three rows a revisited node needs are requested only by the lanes that need them.
        // (The last level a line keeps of a path longer than ring_levels has no successor in the ring: the neighbour row then.)
        const int nxt = lpos + gl + 1;
        const int nl_i = inner ? (nxt < llen ? m.ring_node[line + nxt] : m.nbr[(base + node_i) * kRow + act_i]) : 0;
        const int from = has_pred ? node_i : want;   // where the level above leads: by the same argument the lane's own node, if the level above follows the line
        const u32 rb0 = r.z & 15u, rb1 = (r.z >> 8) & 15u;
        u32 d_i = (arr_i ^ 1u) == rb0 ? rb1 : rb0;
        u64 cnt5 = 0;
        bool again = false, unsure = false;
        for (int j = in_line ? s_head[sel_hash(node_i)] : -1; j >= 0; j = next_at(j)) {   // earlier visits in the chains (levels <= kb)
            if (node_at(j) == node_i) {
                again = true;
                cnt5_add(cnt5, unsure, act_at(j));
                if (j > 0) cnt5_add(cnt5, unsure, act_at(j - 1) ^ 1u);
            }
        }
        const u32 hb = ((u32)node_i * 0x9E3779B1u) >> 25;
        if (in_line) {
            const u32 prev_lane = (u32)atomicExch(&s_seg4[hb], gl);
            s_segent4[gl] = make_uint2((u32)node_i, prev_lane | (act_i << 16) | ((arr_i ^ 1u) << 20));
        }
        __syncthreads();   // every lane of the round is in its bucket's list
        if (in_line) {
            for (u32 j = (u32)s_seg4[hb]; j != 0xFFFFu;) {
                const uint2 en = s_segent4[j];
                if ((int)j < gl && (int)en.x == node_i) {
                    again = true;
                    if (((en.y >> 16) & 15u) != kNoAct) cnt5_add(cnt5, unsure, (en.y >> 16) & 15u);
                    cnt5_add(cnt5, unsure, (en.y >> 20) & 15u);
                }
                j = en.y & 0xFFFFu;
            }
        }
        if (again) {   // the level revisits a node: its decision from its rows, in the lane
            cnt5_add(cnt5, unsure, arr_i ^ 1u);   // own arrival edge
            bool sure;
            const LaneEval e = lane_prepare(c32, m, (base + node_i) * kRow);
            d_i = (u32)lane_pick(e, cnt5, sure);
            unsure |= !sure;
        }
        const bool ok = inner && !(r.z & kRecLeaf) && !(again && unsure) && d_i == act_i && from == node_i;
        const u64 okm = __ballot(ok);
        const int q = ~okm ? __builtin_ctzll(~okm) : kWave;
        if (lane == 0) s_mb[MB_Q + wv] = q;
        if ((int)lane == kWave - 1) { s_mb[MB_LAST_NL + wv] = nl_i; s_mb[MB_LAST_ACT + wv] = (int)act_i; }
        if (q > 0 && (int)lane == q - 1) { s_mb[MB_STOP_NL + wv] = nl_i; s_mb[MB_STOP_ACT + wv] = (int)act_i; }
        if ((int)lane == q) {   // (q < 64) the first level of this wave that does not follow the line
            s_mb[MB_STOP_NODE + wv] = in_line ? node_i : -1;
            s_mb[MB_STOP_REC + 4 * wv] = (int)r.x, s_mb[MB_STOP_REC + 4 * wv + 1] = (int)r.y, s_mb[MB_STOP_REC + 4 * wv + 2] = (int)r.z,
            s_mb[MB_STOP_REC + 4 * wv + 3] = (int)r.w;
        }
        __syncthreads();   // every wave's verdict is in the mailbox, and nobody reads a bucket list any more
        if (in_line) s_seg4[hb] = 0xFFFF;   // empty again for the next round (whose entries come after the next command's barrier)
        int total = 0;
#pragma unroll
        for (int w = 0; w < LW; ++w) {
            const int qw = s_mb[MB_Q + w];
            total += qw;
            if (qw < kWave) break;
        }
        if (gl < total) {   // the leading run of levels that follow the line is appended
            const int kk = kb + 1 + gl;
            put_node(kk, node_i);
            put_act(kk, act_i);
            put_next(kk, atomicExch(&s_head[sel_hash(node_i)], kk));
        }
        return total;
    };
    if (LW > 1 && tid >= kWave) {   // helper waves: rounds on command
        const TreeBufs tbh = tree_bufs(m, base);
        const float c32h = (float)c;
        for (int i = (int)tid - kWave; i < 128; i += (LW - 1) * kWave) s_seg4[i] = 0xFFFF;
        for (;;) {
            __syncthreads();
            const int cmd = s_mb[MB_CMD];
            if (cmd == 0) return;
            if (cmd == 1) line_round(tid >> 6, tid & (kWave - 1), tbh, c32h);
            else __syncthreads();   // pause: every helper has read the 2 before the walking wave may write its next command
        }
    }

    const unsigned long long t_walk = wall_clock64();
    const long long c_walk = clock64();
    int slow_levels = 0, revisits = 0, line_rounds = 0, line_levels = 0;
#ifdef RUBIKS_SELECT_PHASES
    long long cyc_lines = 0;   // shader cycles inside line segments (all their rounds)
    int n_segments = 0, n_loop = 0;
#endif
    const u32 ring_k = m.ring_k;
    const u32 lane = tid;
    const bool act = lane < kA;
    const u32 la = act ? lane : 0;
    const int start = s_first;
    const TreeBufs tb = tree_bufs(m, base);
    const int max_path = DEEP ? path_limit(m, t) : min(path_limit(m, t), kMaxPath);   // levels the path can take right now (shallow body: the LDS window)
    const int ring_levels = (int)m.ring_levels;
    int cur = __builtin_amdgcn_readfirstlane(node_at(start)), plen = start + 1;
    int prev_act = start > 0 ? __builtin_amdgcn_readfirstlane((int)act_at(start - 1)) : -1;   // action that led to `cur`
    u32 walked = 0;
    int stop = 0;
    const float c32 = (float)c;
    u32x4 x = load_rec(tb, cur);
    u32 h = sel_hash(cur);
    int head = s_head[h];
    for (;;) {
#ifdef RUBIKS_SELECT_PHASES
        ++n_loop;
#endif
        const int k = plen - 1;
        // The revisit test of this level needs the path only, so it runs while the node's record (requested a level ago) is
        // still in flight; a revisited node needs its rows for a full evaluation, and they are requested here, next to the
        // record instead of after it: loops through transpositions cost one memory round trip per level, not two.
        put_node(k, cur);   // every lane stores the same value: no exec-mask detour
        u32 cnt = (prev_act >= 0 && (u32)(prev_act ^ 1) == lane) ? 1u : 0u;   // own arrival: L[cur, rev(a_prev)] += nu (agents.py:591)
        bool visited = false;
        for (int j = head; j >= 0; j = next_at(j)) {   // earlier visits of `cur` in this descent: departure edge a_j, arrival edge rev(a_{j-1})
            if (node_at(j) == cur) {
                visited = true;
                cnt += act_at(j) == lane;
                if (j > 0) cnt += (act_at(j - 1) ^ 1u) == lane;
            }
        }
        NodeRows rows_cur = {0, 0, 0.f, 0.f};
        if (visited) rows_cur = load_rows(tb, cur, la);   // wave-uniform
        const u32 z = (u32)__builtin_amdgcn_readfirstlane((int)x.z);
        const int b0 = (int)(z & 15u), b1 = (int)((z >> 8) & 15u);
        const bool back = prev_act >= 0 && (prev_act ^ 1) == b0;   // arrived through rev(best0): that edge carries a loss
        const int a_f = back ? b1 : b0;
        const int next_f = __builtin_amdgcn_readfirstlane((int)(back ? x.y : x.x));
        // the next record is requested at once (a revisited node decides otherwise often enough, but not always)
        u32x4 y = load_rec(tb, next_f);
        u32 hn = sel_hash(next_f);
        int headn = s_head[hn];
        // one branch for the three ways a walk ends; which one is sorted out after the loop
        stop = (z & kRecLeaf) ? 1 : plen >= max_path ? 2 : (level_budget && walked >= level_budget) ? 3 : 0;
        if (stop) break;
        ++walked;
        int arg = a_f, next = next_f;
        if (visited) {   // wave-uniform: full evaluation with the exact loss counts
            const NodeRows r = rows_cur;
            arg = puct_argmax_walk(c, c32, r.n_a, r.p_f, r.w_f, cnt, act, (int)lane, slow_levels);
            next = __builtin_amdgcn_readlane(r.nb, arg);
            ++revisits;
        }
        put_next(k, head);   // the memory operations of one wave are ordered: later levels see this entry
        s_head[h] = k;
        put_act(k, (u32)arg);
        // Line following.  If this node lay on one of the tree's last ring_k descent paths and left it by the same
        // action, the levels that followed it there are the likely continuation: up to 64 of them are checked at
        // once, one lane per level, under the premise that the levels above in the segment follow the line too.  A
        // level passes if its decision for the line's arrival edge is the line's action and it is the node the level
        // above leads to.  The decision comes from the node's record if the node is new to this descent, else from a
        // float32 evaluation of its rows in the lane (lane_pick) with the exact loss counts: earlier visits in the
        // chains plus earlier lanes of the segment at the same node.  The leading run of passing levels is appended
        // in one step: two or three memory round trips per segment instead of one per level.
        // Consecutive segments of one line are chained without returning to the sequential step, and the line's nodes of
        // the NEXT segment are requested while this one is being checked.
        const u32 tag = (u32)__builtin_amdgcn_readfirstlane((int)x.w);
        const u32 tseq = tag >> 16, age = (seq - tseq) & 0xFFFFu;
        int room = max_path - plen - 1;
        if (level_budget) room = min(room, (int)level_budget - (int)walked);
        if (!visited && tseq != 0 && age >= 1 && age <= ring_k && (tag & 15u) == (u32)arg && room > 0) {
            const size_t line = ((size_t)t * ring_k + (tseq & (ring_k - 1))) * (size_t)ring_levels;
            const int llen = m.ring_len[(size_t)t * ring_k + (tseq & (ring_k - 1))];
            int lpos = (int)((tag >> 4) & 0xFFFu) + 1;   // line position that lane 0 checks
            int kb = k;                                   // lane i checks level kb + 1 + i
            int want = next;                              // the node lane 0 must be at
            int arr0 = arg;                               // the action that leads to it
            int total = 0;
            bool have = false;
#ifdef RUBIKS_SELECT_PHASES
            const long long c_seg = clock64();
#endif
            if (LW > 1) {
                for (;;) {
                    if (lane == 0) {
                        s_mb[MB_LINE_LO] = (int)(u32)line, s_mb[MB_LINE_HI] = (int)(u32)(line >> 32), s_mb[MB_LLEN] = llen, s_mb[MB_LPOS] = lpos;
                        s_mb[MB_KB] = kb, s_mb[MB_WANT] = want, s_mb[MB_ARR0] = arr0, s_mb[MB_ROOM] = room, s_mb[MB_CMD] = 1;
                    }
                    __syncthreads();   // command: a round
                    const int got = line_round(0, lane, tb, c32);
                    ++line_rounds;
                    total += got;
                    if (got > 0) {
                        want = s_mb[MB_STOP_NL + (got - 1) / kWave];
                        arr0 = s_mb[MB_STOP_ACT + (got - 1) / kWave];
                    }
                    if (got < LW * kWave) {   // the run ends inside this round: the record of the level it ends at, if that is the node we are at
                        const int ws = got / kWave;
                        have = s_mb[MB_STOP_NODE + ws] == want;
                        if (have) {
                            x.x = (u32)s_mb[MB_STOP_REC + 4 * ws], x.y = (u32)s_mb[MB_STOP_REC + 4 * ws + 1];
                            x.z = (u32)s_mb[MB_STOP_REC + 4 * ws + 2], x.w = (u32)s_mb[MB_STOP_REC + 4 * ws + 3];
                        }
                        break;
                    }
                    kb += LW * kWave;
                    lpos += LW * kWave;
                    room -= LW * kWave;
                    if (lpos >= llen || room <= 0) break;   // line or room exhausted
                }
                if (lane == 0) s_mb[MB_CMD] = 2;
                __syncthreads();   // command: pause -- the levels the helpers appended are in the chains
                __syncthreads();   // ... and the helpers have read it (they answer a pause with this barrier): MB_CMD may change again
            }
            bool in_line = LW == 1 && lpos + (int)lane < llen && (int)lane < room;
            int node_i = in_line ? m.ring_node[line + lpos + lane] : 0;
            u32 act_i = in_line ? (u32)m.ring_act[line + lpos + lane] : kNoAct;
            u32 arr_i = (lane == 0 || !in_line) ? (u32)arr0 : (u32)m.ring_act[line + lpos + lane - 1];
            for (; LW == 1;) {
                const u32x4 r = load_rec(tb, node_i);
                const bool inner = in_line && act_i != kNoAct;
                const int nx1 = lpos + (int)lane + 1;   // where the line's action leads: the line's next node (see line_round)
                const int nl_i = inner ? (nx1 < llen ? m.ring_node[line + nx1] : m.nbr[(base + node_i) * kRow + act_i]) : 0;
                // the next segment of the line, in flight while this one is checked
                const int li2 = lpos + kWave + (int)lane;
                const bool in2 = li2 < llen && kWave + (int)lane < room;
                const int node2 = in2 ? m.ring_node[line + li2] : 0;
                const u32 act2 = in2 ? (u32)m.ring_act[line + li2] : kNoAct;
                const u32 arr2 = in2 ? (u32)m.ring_act[line + li2 - 1] : 0u;
                const u32 rb0 = r.z & 15u, rb1 = (r.z >> 8) & 15u;
                u32 d_i = (arr_i ^ 1u) == rb0 ? rb1 : rb0;
                // earlier visits of the lane's node: in the chains (levels <= kb) ...
                u64 cnt5 = 0;
                bool again = false, unsure = false;
                for (int j = in_line ? s_head[sel_hash(node_i)] : -1; j >= 0; j = next_at(j)) {
                    if (node_at(j) == node_i) {
                        again = true;
                        cnt5_add(cnt5, unsure, act_at(j));
                        if (j > 0) cnt5_add(cnt5, unsure, act_at(j - 1) ^ 1u);
                    }
                }
                // ... and among the lanes above (they follow the line by premise: departure act_j, arrival rev(arr_j)): the lanes
                // of the segment are chained by node bucket in LDS, and every lane walks its bucket's (short) list.
                for (int i = lane; i < 256; i += kWave) s_seg[i] = 0xFF;
                const u32 hb = ((u32)node_i * 0x9E3779B1u) >> 24;
                if (in_line) {
                    const u32 prev_lane = (u32)atomicExch(&s_seg[hb], (int)lane);
                    s_segent[lane] = make_uint2((u32)node_i, prev_lane | (act_i << 8) | ((arr_i ^ 1u) << 12));
                    for (u32 j = (u32)s_seg[hb]; j != 0xFFu;) {
                        const uint2 en = s_segent[j];
                        if (j < lane && (int)en.x == node_i) {
                            again = true;
                            if (((en.y >> 8) & 15u) != kNoAct) cnt5_add(cnt5, unsure, (en.y >> 8) & 15u);
                            cnt5_add(cnt5, unsure, (en.y >> 12) & 15u);
                        }
                        j = en.y & 0xFFu;
                    }
                }
                if (again) {   // the level revisits a node (few lanes do): its decision from its rows, requested here, in the lane
                    cnt5_add(cnt5, unsure, arr_i ^ 1u);   // own arrival edge
                    bool sure;
                    const LaneEval e = lane_prepare(c32, m, (base + node_i) * kRow);
                    d_i = (u32)lane_pick(e, cnt5, sure);
                    unsure |= !sure;
                }
                const int from = __shfl_up(nl_i, 1);
                const bool ok = inner && !(r.z & kRecLeaf) && !(again && unsure) && d_i == act_i && (lane == 0 ? want : from) == node_i;
                const u64 okm = __ballot(ok);
                const int q = ~okm ? __builtin_ctzll(~okm) : kWave;
                ++line_rounds;
                revisits += __popcll(__ballot(again) & ((q < kWave ? (1ull << q) : 0ull) - 1ull));
                if ((int)lane < q) {
                    const int kk = kb + 1 + (int)lane;
                    put_node(kk, node_i);
                    put_act(kk, act_i);
                    put_next(kk, atomicExch(&s_head[sel_hash(node_i)], kk));
                }
                total += q;
                if (q > 0) {
                    want = __builtin_amdgcn_readlane(nl_i, q - 1);
                    arr0 = (int)__builtin_amdgcn_readlane((int)act_i, q - 1);
                }
                if (q < kWave) {   // the run ends inside this segment: lane q's record, if it is the node we are at, is at hand
                    have = (__ballot(in_line && node_i == want) >> q) & 1ull;
                    if (have) {
                        x.x = (u32)__builtin_amdgcn_readlane((int)r.x, q);
                        x.y = (u32)__builtin_amdgcn_readlane((int)r.y, q);
                        x.z = (u32)__builtin_amdgcn_readlane((int)r.z, q);
                        x.w = (u32)__builtin_amdgcn_readlane((int)r.w, q);
                    }
                    break;
                }
                kb += kWave;
                lpos += kWave;
                room -= kWave;
                in_line = in2;
                node_i = node2;
                act_i = act2;
                arr_i = arr2;
                if (!__ballot(in_line)) break;   // line or room exhausted
            }
#ifdef RUBIKS_SELECT_PHASES
            cyc_lines += clock64() - c_seg;
            ++n_segments;
#endif
            if (total > 0) {
                cur = want;
                prev_act = arr0;
                if (!have) x = load_rec(tb, cur);
                h = sel_hash(cur);
                head = s_head[h];
                plen += 1 + total;
                walked += (u32)total;
                line_levels += total;
                continue;
            }
        }
        if (next != next_f) {   // wave-uniform
            y = load_rec(tb, next);
            hn = sel_hash(next);
            headn = s_head[hn];   // after the insertion above, so a colliding bucket is seen complete
        } else if (hn == h) {
            headn = k;            // ... as here, where the early read missed it
        }
        x = y;
        h = hn;
        head = headn;
        prev_act = arg;
        cur = next;
        ++plen;
    }
    if (LW > 1) {
        if (lane == 0) s_mb[MB_CMD] = 0;
        __syncthreads();   // command: the helpers leave
    }
#ifdef RUBIKS_SELECT_PHASES
    const unsigned long long ph4 = wall_clock64();   // end of the walk loop
#endif
    // The path store is full.  If that is the memory behind this tree's path blocks (path_rows), the descent is suspended like one
    // that ran out of its level budget: the host maps the next block, a later call resumes.  Only the end of the store itself ends
    // the tree (the reference has no such limit, agents.py:575-595: max_path is the caller's resource bound).
    if (stop == 2 && max_path >= (int)m.max_path && lane == 0) m.status[t] = RC_MCTS_PATH_OVERFLOW;
    const int suspended = stop == 3 || (stop == 2 && max_path < (int)m.max_path);   // resume here next call
    // the walked levels go to memory (the path carries their virtual losses, agents.py:589-591); the deep ones are there already
    for (int k = start + (int)lane; k < min(plen, W); k += kWave) {
        const size_t pk = P(k);
        if (k > start) m.path_node[pk] = s_node[k];
        if (k < plen - 1) m.path_act[pk] = (u8)s_act[k];
    }
#ifdef RUBIKS_SELECT_PHASES
    const unsigned long long ph5 = wall_clock64();   // walked levels written back
#endif
    // the path's first ring_levels levels become line `seq` of the ring, and every node on them is tagged with its place there
    if (seq != 0) {
        const size_t slot = (size_t)t * ring_k + (seq & (ring_k - 1));
        u32 *rec_w = reinterpret_cast<u32 *>(m.rec) + base * kRow + 3;
        const int llen = min(plen, ring_levels);
        for (int k = (int)lane; k < llen; k += kWave) {
            const u32 a = k < plen - 1 ? act_at(k) : kNoAct;
            const int nk = node_at(k);
            m.ring_node[slot * ring_levels + k] = nk;
            m.ring_act[slot * ring_levels + k] = (u8)a;
            rec_w[(size_t)nk * kRow] = line_tag(seq, k, a);
        }
        if (lane == 0) m.ring_len[slot] = llen;
    }
    if (lane == 0) {
        if (m.select_stats) {
            m.select_stats[8 * t] = start;
            m.select_stats[8 * t + 1] = plen;
            m.select_stats[8 * t + 2] = (int)(t_walk - t_begin);            // 10 ns ticks: staging + re-validation
            m.select_stats[8 * t + 3] = (int)(wall_clock64() - t_walk);     // ... sequential walk
            m.select_stats[8 * t + 4] = (int)(clock64() - c_walk);          // shader cycles of the walk
            // low half: float64 fall-backs of this walk; high half: the most levels pass A has left to pass B in any call so far
            const int unc_now = resume ? 0 : min(s_nunc, 0x7FFF), unc_max = max(m.select_stats[8 * t + 5] >> 16, unc_now);
            m.select_stats[8 * t + 5] = (unc_max << 16) | min(slow_levels, 0xFFFF);
            m.select_stats[8 * t + 6] = revisits;
            m.select_stats[8 * t + 7] = (line_rounds << 16) | min(line_levels, 0xFFFF);
#ifdef RUBIKS_SELECT_PHASES
            m.select_stats[8 * t + 5] = (int)(ph1 - t_begin), m.select_stats[8 * t + 6] = (int)(ph2 - ph1), m.select_stats[8 * t + 7] = (int)(ph3 - ph2);
            if (RUBIKS_SELECT_PHASES == 2) m.select_stats[8 * t + 6] = (int)(ph3 - ph1), m.select_stats[8 * t + 7] = s_nlate;   // first levels; number of late levels
            if (RUBIKS_SELECT_PHASES == 4)   // shader cycles inside line segments | segments | passes of the walk loop
                m.select_stats[8 * t + 5] = (int)cyc_lines, m.select_stats[8 * t + 6] = n_segments, m.select_stats[8 * t + 7] = n_loop;
            if (RUBIKS_SELECT_PHASES == 3)   // the walk: its loop | write-back of the walked levels | ring line + tags
                m.select_stats[8 * t + 5] = (int)(ph4 - t_walk), m.select_stats[8 * t + 6] = (int)(ph5 - ph4), m.select_stats[8 * t + 7] = (int)(wall_clock64() - ph5);
#endif
        }
        m.path_len[t] = plen;
        m.pending[t] = suspended;   // 1 = resume at path_len - 1
    }
    if (FUSE && stop == 1)   // the descent stands on a leaf: its expansion, the first act of the next iteration (agents.py:477)
        expand_leaf_wave(m, t, slot, lane, reinterpret_cast<const u8 *>(s_lut), max_states, false, cur);
    };
    if ((int)m.lds_levels < kMaxPath || plen_old >= kMaxPath) body(std::true_type{});
    else body(std::false_type{});
}

// ---- _complete_graph (agents.py:597-611): one workgroup per solved tree, one thread per (leaf, action) ----
__global__ __launch_bounds__(kBlock) void k_mcts_complete_graph(rc_mcts_t m) {
    __shared__ u32 s_lut[sizeof(kTables.lut) / 4];
    stage_to_lds(s_lut, c_tables.lut, sizeof(kTables.lut));
    __syncthreads();
    const u8 *lut = reinterpret_cast<const u8 *>(s_lut);
    const int ti = tree_of(m, blockIdx.x);
    if (ti < 0) return;
    const u32 t = (u32)ti;
    if (m.status[t] != RC_MCTS_SOLVED) return;
    const size_t base = (size_t)t * (m.capacity + 1);
    const uint4 *keys = reinterpret_cast<const uint4 *>(m.keys) + base;
    const int *tab = m.hash + (size_t)t * m.hash_size;
    const u32 mask = m.hash_size - 1;
    const u32 n = (u32)m.n_nodes[t];
    // gridDim.y workgroups share a tree: the (leaf, action) pairs are independent, and a tree finished at 50 000 nodes
    // would otherwise keep one CU busy for milliseconds next to the lock-step iterations of the running trees
    for (u32 w = blockIdx.y * kBlock + threadIdx.x; w < n * kA; w += kBlock * gridDim.y) {
        const u32 node = 1 + w / kA, a = w % kA;
        if (!m.leaf[base + node]) continue;
        const uint4 pk = keys[node];
        u32 kw[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < kPlanes; ++j)
            key_set(kw, j, lut[a * (2 * kCodePad) + (j >= kCorners ? kCodePad : 0) + key_code(pk, j)]);
        const uint4 ck = make_uint4(kw[0], kw[1], kw[2], kw[3]);
        u32 h = key_hash(ck) & mask;
        int found = 0;
        for (u32 probes = 0;; ++probes) {
            const int s = tab[h];
            if (s == 0) break;
            // A hash slot names a node of this tree.  Anything else is not this tree's data (a copy that went through a stale
            // translation, a harvest of the wrong rows): the tree is marked and left alone instead of being indexed with it.
            if ((u32)s > n || probes > mask) { m.status[t] = RC_MCTS_CORRUPT; return; }
            if (key_eq(keys[s], ck)) { found = s; break; }
            h = (h + 1) & mask;
        }
        m.nbr[(base + node) * m.node_words + a] = found;                // 0 when the child is not in the tree (results-only forests: 12 words per node)
        if (found) m.nbr[(base + found) * m.node_words + (a ^ 1)] = (int)node;  // the only node whose action a^1 leads here
    }
}

// ---- _shorten_action_queue (agents.py:613-633): ordered BFS, one workgroup per solved tree --------------
// Level-synchronous, but every node keeps the discoverer the reference's FIFO scan would give it: candidate
// (frontier position i, action a) carries the scan index 12 i + a, the smallest index claims the node
// (atomicMin), and the next frontier is the claimed nodes in scan-index order (block-wide ordered compaction).
__global__ __launch_bounds__(kBlock) void k_mcts_shorten(rc_mcts_t m) {
    __shared__ int s_scan[kBlock];
    __shared__ int s_base, s_done, s_bad;
    const u32 tid = threadIdx.x;
    const int ti = tree_of(m, blockIdx.x);
    if (ti < 0) return;
    const u32 t = (u32)ti;
    if (tid == 0) m.short_len[t] = -1;
    if (m.status[t] != RC_MCTS_SOLVED) return;
    const int solved = m.solved_idx[t];
    if (solved == 1) return;   // agents.py:614: the queue is kept
    const size_t base = (size_t)t * (m.capacity + 1);
    const int n = m.n_nodes[t];
    // Every index this kernel follows is read from memory (neighbour rows, parent links).  One that does not name a node of this
    // tree (1 .. n) means the rows are not this tree's data: the tree is marked RC_MCTS_CORRUPT and left alone -- a diagnosable
    // error on the host instead of a wild access.
    if (n < 1 || (u32)n > m.capacity || solved < 1 || solved > n) {
        if (tid == 0) m.status[t] = RC_MCTS_CORRUPT;
        return;
    }
    const u32 rw = m.node_words;   // 64 in a search forest, 12 in a results-only one
    const int *nbr = m.nbr + base * rw;
    int *claim = m.bfs + base * 2;            // [node][0]
    int *frontier_a = m.hash + (size_t)t * m.hash_size, *frontier_b = frontier_a + (m.capacity + 1);
    for (int i = tid; i <= n; i += kBlock) {
        claim[2 * i] = INT_MAX;
        claim[2 * i + 1] = 0;   // parent << 4 | action; 0 = not visited
    }
    if (tid == 0) {
        frontier_a[0] = 1;
        s_done = 0;
        s_bad = 0;
    }
    __syncthreads();
    if (tid == 0) claim[2 * 1 + 1] = -1;   // the root is visited and has no parent
    int fsize = 1;
    int *cur = frontier_a, *nxt = frontier_b;
    while (fsize > 0) {
        __syncthreads();
        const int work = fsize * kA;
        // phase 1: every unvisited neighbour is claimed by the smallest scan index that reaches it
        for (int idx = tid; idx < work; idx += kBlock) {
            const int c = nbr[(size_t)cur[idx / kA] * rw + idx % kA];
            if ((u32)c > (u32)n) { s_bad = 1; continue; }
            if (c != 0 && claim[2 * c + 1] == 0) atomicMin(&claim[2 * c], idx);
        }
        // the claims are no-return atomics executed at L2: drain them before the barrier, and read them back
        // with L2-served loads (a plain load could hit a stale L1 line)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // phase 2: winners enter the next frontier in scan order
        if (tid == 0) s_base = 0;
        __syncthreads();
        if (s_bad) break;   // uniform: written before the barrier above
        for (int idx0 = 0; idx0 < work; idx0 += kBlock) {
            const int idx = idx0 + (int)tid;
            int c = 0, win = 0, p = 0;
            if (idx < work) {
                p = cur[idx / kA];
                c = nbr[(size_t)p * rw + idx % kA];
                win = (c != 0 && claim[2 * c + 1] == 0 &&
                       __hip_atomic_load(&claim[2 * c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == idx) ? 1 : 0;
            }
            s_scan[tid] = win;
            __syncthreads();
            for (int o = 1; o < kBlock; o <<= 1) {
                const int x = (tid >= (u32)o) ? s_scan[tid - o] : 0;
                __syncthreads();
                s_scan[tid] += x;
                __syncthreads();
            }
            const int pos = s_base + s_scan[tid] - win;
            if (win) {
                nxt[pos] = c;
                claim[2 * c + 1] = (p << 4) | (idx % kA);
                if (c == solved) s_done = 1;
            }
            __syncthreads();
            if (tid == kBlock - 1) s_base += s_scan[tid];
            __syncthreads();
        }
        fsize = s_base;
        if (s_done) break;
        int *tmp = cur; cur = nxt; nxt = tmp;
    }
    __syncthreads();
    if (tid == 0 && s_bad) m.status[t] = RC_MCTS_CORRUPT;
    if (tid == 0 && s_done && !s_bad) {   // walk the parent pointers back to the root, then reverse
        // (the shortest path is never longer than the tree's last descent path + its solving move, whose levels have memory)
        u8 *out = m.short_act;
        const int cap = path_limit(m, t);
        int len = 0;
        for (int v = solved; v != 1 && len < cap; v = claim[2 * v + 1] >> 4) out[path_at(m, t, len++)] = (u8)(claim[2 * v + 1] & 15);
        for (int i = 0; i < len / 2; ++i) {
            const size_t pa = path_at(m, t, i), pb = path_at(m, t, len - 1 - i);
            const u8 x = out[pa];
            out[pa] = out[pb];
            out[pb] = x;
        }
        m.short_len[t] = len;
    }
}

// ---- finished trees leave a forest: rows 0 .. n_nodes of what the destination keeps, and the tree's hash table ---------
// grid (trees, parts): the parts of a tree's workgroups stride over its rows.  FULL: into a search forest (records + V),
// else into a results-only forest (neighbour rows as a plain [rows][12] array).
// HASH: the tree's hash table travels as it is (source and destination have the same capacity); otherwise the destination's table --
// of its own size -- is rebuilt from the copied keys by k_mcts_rehash.
template <bool FULL, bool HASH>
__global__ __launch_bounds__(kBlock) void k_mcts_copy_trees(rc_mcts_t src, rc_mcts_t dst, const int *__restrict__ src_trees, u32 dst_first) {
    const u32 ts = (u32)src_trees[blockIdx.x], td = dst_first + blockIdx.x;
    const size_t sb = (size_t)ts * (src.capacity + 1), db = (size_t)td * (dst.capacity + 1);
    const u32 rows = (u32)src.n_nodes[ts] + 1u;
    const u32 tid = blockIdx.y * kBlock + threadIdx.x, nt = gridDim.y * kBlock;
    const uint4 *skeys = reinterpret_cast<const uint4 *>(src.keys) + sb;
    uint4 *dkeys = reinterpret_cast<uint4 *>(dst.keys) + db;
    for (u32 i = tid; i < rows; i += nt) {
        dkeys[i] = skeys[i];
        dst.leaf[db + i] = src.leaf[sb + i];
        if (FULL) dst.V[db + i] = src.V[sb + i];
    }
    if (FULL) {   // the whole 256-byte record: 16 lanes per row
        const uint4 *srec = reinterpret_cast<const uint4 *>(src.N) + sb * (kRow / 4);
        uint4 *drec = reinterpret_cast<uint4 *>(dst.N) + db * (kRow / 4);
        for (size_t i = tid; i < (size_t)rows * (kRow / 4); i += nt) drec[i] = srec[i];
    } else {      // words 44 .. 55 of the record -> row of 12
        const uint4 *snbr = reinterpret_cast<const uint4 *>(src.nbr);
        uint4 *dnbr = reinterpret_cast<uint4 *>(dst.nbr);
        for (u32 i = tid; i < rows * 3u; i += nt) dnbr[(db + i / 3u) * 3u + i % 3u] = snbr[(sb + i / 3u) * (kRow / 4) + i % 3u];
    }
    if (HASH) {
        const uint4 *shash = reinterpret_cast<const uint4 *>(src.hash + (size_t)ts * src.hash_size);
        uint4 *dhash = reinterpret_cast<uint4 *>(dst.hash + (size_t)td * dst.hash_size);
        for (u32 i = tid; i < src.hash_size / 4; i += nt) dhash[i] = shash[i];
    }
}

// The hash table of trees dst_first .. of a forest, rebuilt from their keys (nodes 1 .. n_nodes, which the caller has just copied in from a
// forest of another capacity): cleared, then every key claims the first free slot from its home (the order does not matter to a lookup).
// One workgroup per tree.
__global__ __launch_bounds__(kBlock) void k_mcts_rehash(rc_mcts_t m, const rc_mcts_t src, const int *__restrict__ src_trees, u32 dst_first) {
    const u32 t = dst_first + blockIdx.x;
    const int n = src.n_nodes[src_trees[blockIdx.x]];   // (the destination's per-tree words are copied by the host afterwards)
    int *tab = m.hash + (size_t)t * m.hash_size;
    uint4 *tab4 = reinterpret_cast<uint4 *>(tab);
    for (u32 i = threadIdx.x; i < m.hash_size / 4; i += kBlock) tab4[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const uint4 *keys = reinterpret_cast<const uint4 *>(m.keys) + (size_t)t * (m.capacity + 1);
    const u32 mask = m.hash_size - 1;
    for (int i = 1 + (int)threadIdx.x; i <= n; i += kBlock) {
        u32 h = key_hash(keys[i]) & mask;
        while (atomicCAS(&tab[h], 0, i) != 0) h = (h + 1) & mask;
    }
}

}  // namespace rubiks

using namespace rubiks;

static inline unsigned mcts_grid(const rc_mcts_t *m) { return m->active ? m->n_active : m->n_trees; }

// Threads per tree of the one-launch iteration: 256 in a full forest; 512 / 1 024 where the forest is small enough to leave CUs
// idle (RUBIKS_STEP_THREADS=256 pins it, for A/B measurements).
// Waves that check a line together in the one-launch iteration: four where the forest is small enough to leave CUs idle (runs to
// completion -1 ... -3 %), one in a full forest, whose 16 waves per CU gain nothing from it (RUBIKS_LINE_WAVES=1 / 4 pins it, for
// A/B measurements; profiles/r3_line_waves_ab.txt).
static int line_waves(unsigned n_trees) {
    static const int pinned = [] { const char *e = getenv("RUBIKS_LINE_WAVES"); return e ? atoi(e) : 0; }();
    return pinned == 4 ? 4 : pinned == 1 ? 1 : n_trees <= 512 ? 4 : 1;
}
static unsigned step_threads(unsigned n_trees) {
    static const int pinned = [] { const char *e = getenv("RUBIKS_STEP_THREADS"); return e ? atoi(e) : 0; }();
    if (pinned == 256 || pinned == 512 || pinned == 1024) return (unsigned)pinned;
    return n_trees <= 256 ? 1024u : n_trees <= 512 ? 512u : 256u;
}

static int check_mcts(const rc_mcts_t *m, bool results_only_ok = false) {
    RC_REQUIRE(m != nullptr, RC_ERR_NULL);
    RC_REQUIRE(m->keys && m->nbr && m->P && m->W && m->N && m->V && m->leaf && m->hash && m->n_nodes &&
                   m->status && m->solved_idx && m->solved_action && m->iterations && m->path_len && m->path_node &&
                   m->pending && m->path_act && m->child_soa && m->child_idx && m->new_mask && m->expanded && m->rec && m->ring_node && m->ring_act &&
                   m->ring_len && m->phase,
               RC_ERR_NULL);
    RC_REQUIRE(m->n_trees > 0 && m->capacity >= 13 && m->max_path >= 2 && m->max_path < (1u << 30), RC_ERR_RANGE);
    // the blocked path arrays: whole blocks; the first lds_levels levels of a path are worked on in LDS, the first ring_levels kept as lines
    RC_REQUIRE(m->path_block_log2 >= 1 && m->path_block_log2 <= 24 && (m->max_path & ((1u << m->path_block_log2) - 1u)) == 0, RC_ERR_RANGE);
    RC_REQUIRE(m->lds_levels >= 1 && m->lds_levels <= (uint32_t)kMaxPath && m->ring_levels >= 1 && m->ring_levels <= (uint32_t)kMaxPath, RC_ERR_RANGE);
    RC_REQUIRE(results_only_ok || m->path_next != nullptr, RC_ERR_NULL);
    // the kernels address a tree's node records through 32-bit buffer resources: (capacity + 1) * 256 bytes must stay below 2^32
    RC_REQUIRE(m->capacity + 1 < (1u << 24), RC_ERR_RANGE);   // a tree's node records are addressed by 32-bit byte offsets (tree_bufs)
    RC_REQUIRE(m->active == nullptr || (m->n_active >= 1 && m->n_active <= m->n_trees), RC_ERR_RANGE);
    RC_REQUIRE((m->hash_size & (m->hash_size - 1)) == 0 && m->hash_size >= 2 * (m->capacity + 1), RC_ERR_RANGE);
    RC_REQUIRE(aligned16(m->keys) && aligned16(m->rec) && aligned16(m->child_soa) && (m->child_stride & 15u) == 0, RC_ERR_ALIGN);
    if (results_only_ok && m->node_words == (uint32_t)kA) {
        // a results-only forest (rc_mcts_complete_graph / rc_mcts_shorten): nbr is a plain [rows][12] array, the other per-action
        // pointers are not read
    } else {   // the fields of the node record sit where the kernels' strides assume them (one allocation, 256-byte aligned)
        RC_REQUIRE(m->node_words == (uint32_t)kRow, RC_ERR_RANGE);
        const uintptr_t n0 = reinterpret_cast<uintptr_t>(m->N);
        RC_REQUIRE((n0 & 255u) == 0 && reinterpret_cast<uintptr_t>(m->W) == n0 + 48 && reinterpret_cast<uintptr_t>(m->rec) == n0 + 96 &&
                       reinterpret_cast<uintptr_t>(m->P) == n0 + 128 && reinterpret_cast<uintptr_t>(m->nbr) == n0 + 176, RC_ERR_ALIGN);
    }
    RC_REQUIRE(m->rows_per_tree == 11, RC_ERR_RANGE);
    RC_REQUIRE(m->ring_k >= 1 && m->ring_k <= 64 && (m->ring_k & (m->ring_k - 1)) == 0, RC_ERR_RANGE);
    RC_REQUIRE(m->child_stride >= round_up((size_t)mcts_grid(m) * m->rows_per_tree, 16), RC_ERR_STRIDE);
    return RC_OK;
}

extern "C" {

size_t rc_mcts_struct_bytes(void) { return sizeof(rc_mcts_t); }

int rc_mcts_plant(const rc_mcts_t *m, const int32_t *slots, uint32_t n_slots, const int8_t *roots_soa, size_t stride, size_t first_col,
                  rc_stream_t stream) {
    if (int rc = check_mcts(m)) return rc;
    if (n_slots == 0) return RC_OK;
    RC_REQUIRE(roots_soa != nullptr, RC_ERR_NULL);
    RC_REQUIRE(aligned16(roots_soa) && (stride & 15u) == 0, RC_ERR_ALIGN);
    RC_REQUIRE(n_slots <= m->n_trees && stride >= first_col + n_slots && (m->hash_size & 3u) == 0, RC_ERR_RANGE);
    hipLaunchKernelGGL(k_mcts_plant, dim3(n_slots), dim3(kBlock), 0, (hipStream_t)stream, *m, (const int *)slots, (const u8 *)roots_soa,
                       stride, first_col, 0u);
    return launch_status();
}

int rc_mcts_plant_expanded(const rc_mcts_t *m, const int32_t *slots, uint32_t n_slots, const int8_t *roots_soa, size_t stride,
                           size_t first_col, uint32_t max_states, rc_stream_t stream) {
    if (int rc = check_mcts(m)) return rc;
    if (n_slots == 0) return RC_OK;
    RC_REQUIRE(roots_soa != nullptr, RC_ERR_NULL);
    RC_REQUIRE(aligned16(roots_soa) && (stride & 15u) == 0, RC_ERR_ALIGN);
    RC_REQUIRE(n_slots <= m->n_trees && stride >= first_col + n_slots && (m->hash_size & 3u) == 0 && max_states > 0, RC_ERR_RANGE);
    // the root's network rows are those of list position == tree index: every tree must be listed, in order
    RC_REQUIRE(m->active == nullptr || m->n_active == m->n_trees, RC_ERR_RANGE);
    hipLaunchKernelGGL(k_mcts_plant, dim3(n_slots), dim3(kBlock), 0, (hipStream_t)stream, *m, (const int *)slots, (const u8 *)roots_soa,
                       stride, first_col, max_states);
    return launch_status();
}

int rc_mcts_expand(const rc_mcts_t *m, uint32_t max_states, rc_stream_t stream) {
    if (int rc = check_mcts(m)) return rc;
    hipLaunchKernelGGL(k_mcts_expand, dim3(mcts_grid(m)), dim3(kWave), 0, (hipStream_t)stream, *m, max_states);
    return launch_status();
}

int rc_mcts_backup(const rc_mcts_t *m, const float *probs, const float *values, rc_stream_t stream) {
    if (int rc = check_mcts(m)) return rc;
    RC_REQUIRE(probs && values, RC_ERR_NULL);
    hipLaunchKernelGGL(k_mcts_backup<false>, dim3(mcts_grid(m)), dim3(kBlock), 0, (hipStream_t)stream, *m, (const void *)probs,
                       values, (size_t)0, false);
    return launch_status();
}

int rc_mcts_backup_head(const rc_mcts_t *m, const void *head, size_t ld, int head_is_bf16, rc_stream_t stream) {
    if (int rc = check_mcts(m)) return rc;
    RC_REQUIRE(head != nullptr, RC_ERR_NULL);
    RC_REQUIRE(ld >= (size_t)kActions + 1, RC_ERR_RANGE);
    hipLaunchKernelGGL(k_mcts_backup<true>, dim3(mcts_grid(m)), dim3(kBlock), 0, (hipStream_t)stream, *m, head,
                       (const float *)nullptr, ld, head_is_bf16 != 0);
    return launch_status();
}

int rc_mcts_complete_graph(const rc_mcts_t *m, rc_stream_t stream) {
    if (int rc = check_mcts(m, true)) return rc;
    const unsigned n = mcts_grid(m), split = n >= 512 ? 1 : n >= 64 ? 8 : 32;
    hipLaunchKernelGGL(k_mcts_complete_graph, dim3(n, split), dim3(kBlock), 0, (hipStream_t)stream, *m);
    return launch_status();
}

int rc_mcts_shorten(const rc_mcts_t *m, rc_stream_t stream) {
    if (int rc = check_mcts(m, true)) return rc;
    RC_REQUIRE(m->bfs && m->short_act && m->short_len, RC_ERR_NULL);
    hipLaunchKernelGGL(k_mcts_shorten, dim3(mcts_grid(m)), dim3(kBlock), 0, (hipStream_t)stream, *m);
    return launch_status();
}

int rc_mcts_copy_trees(const rc_mcts_t *src, const rc_mcts_t *dst, const int32_t *src_trees, uint32_t n, uint32_t dst_first,
                       rc_stream_t stream) {
    if (int rc = check_mcts(src)) return rc;
    if (int rc = check_mcts(dst, true)) return rc;
    if (n == 0) return RC_OK;
    RC_REQUIRE(src_trees != nullptr, RC_ERR_NULL);
    RC_REQUIRE(n <= src->n_trees && dst_first <= dst->n_trees && n <= dst->n_trees - dst_first, RC_ERR_RANGE);
    // the destination may be a forest of another (smaller) capacity -- the caller has made sure the trees fit its rows (n_nodes <= capacity):
    // the hash tables then differ in size and the destination's are rebuilt from the keys
    RC_REQUIRE((src->hash_size & 3u) == 0 && (dst->hash_size & 3u) == 0, RC_ERR_RANGE);
    RC_REQUIRE(src->keys != dst->keys, RC_ERR_RANGE);
    const bool same = src->capacity == dst->capacity && src->hash_size == dst->hash_size;
    const unsigned parts = n >= 256 ? 2 : n >= 32 ? 8 : 32;
    const bool full = dst->node_words == (uint32_t)kRow;
#define RC_COPY(FULL_, HASH_) \
    hipLaunchKernelGGL((k_mcts_copy_trees<FULL_, HASH_>), dim3(n, parts), dim3(kBlock), 0, (hipStream_t)stream, *src, *dst, (const int *)src_trees, dst_first)
    if (full && same) RC_COPY(true, true);
    else if (full) RC_COPY(true, false);
    else if (same) RC_COPY(false, true);
    else RC_COPY(false, false);
#undef RC_COPY
    if (!same) hipLaunchKernelGGL(k_mcts_rehash, dim3(n), dim3(kBlock), 0, (hipStream_t)stream, *dst, *src, (const int *)src_trees, dst_first);
    return launch_status();
}

int rc_mcts_select(const rc_mcts_t *m, double c, uint32_t level_budget, rc_stream_t stream) {
    if (int rc = check_mcts(m)) return rc;
    hipLaunchKernelGGL(k_mcts_select<0>, dim3(mcts_grid(m)), dim3(kBlock), 0, (hipStream_t)stream, *m, c, level_budget,
                       (const void *)nullptr, (const float *)nullptr, (size_t)0, false, 0u);
    return launch_status();
}

int rc_mcts_backup_select(const rc_mcts_t *m, const float *probs, const float *values, double c, uint32_t level_budget,
                          rc_stream_t stream) {
    if (int rc = check_mcts(m)) return rc;
    RC_REQUIRE(probs && values, RC_ERR_NULL);
    hipLaunchKernelGGL(k_mcts_select<1>, dim3(mcts_grid(m)), dim3(kBlock), 0, (hipStream_t)stream, *m, c, level_budget,
                       (const void *)probs, values, (size_t)0, false, 0u);
    return launch_status();
}

int rc_mcts_backup_select_head(const rc_mcts_t *m, const void *head, size_t ld, int head_is_bf16, double c, uint32_t level_budget,
                               rc_stream_t stream) {
    if (int rc = check_mcts(m)) return rc;
    RC_REQUIRE(head != nullptr, RC_ERR_NULL);
    RC_REQUIRE(ld >= (size_t)kActions + 1, RC_ERR_RANGE);
    hipLaunchKernelGGL(k_mcts_select<2>, dim3(mcts_grid(m)), dim3(kBlock), 0, (hipStream_t)stream, *m, c, level_budget, head,
                       (const float *)nullptr, ld, head_is_bf16 != 0, 0u);
    return launch_status();
}

int rc_mcts_step(const rc_mcts_t *m, const float *probs, const float *values, double c, uint32_t level_budget, uint32_t max_states,
                 rc_stream_t stream) {
    if (int rc = check_mcts(m)) return rc;
    RC_REQUIRE(probs && values, RC_ERR_NULL);
    RC_REQUIRE(max_states > 0, RC_ERR_RANGE);
    const unsigned g = mcts_grid(m);
#define RC_STEP(NT_)                                                                                                                        \
    do {                                                                                                                                    \
        if (line_waves(g) == 4)                                                                                                              \
            hipLaunchKernelGGL((k_mcts_select<1, true, NT_, 4>), dim3(g), dim3(NT_), 0, (hipStream_t)stream, *m, c, level_budget,            \
                               (const void *)probs, values, (size_t)0, false, max_states);                                                  \
        else                                                                                                                                \
            hipLaunchKernelGGL((k_mcts_select<1, true, NT_, 1>), dim3(g), dim3(NT_), 0, (hipStream_t)stream, *m, c, level_budget,            \
                               (const void *)probs, values, (size_t)0, false, max_states);                                                  \
    } while (0)
    const unsigned nt = step_threads(g);
    if (nt == 1024) RC_STEP(1024);
    else if (nt == 512) RC_STEP(512);
    else RC_STEP(256);
#undef RC_STEP
    return launch_status();
}

int rc_mcts_step_head(const rc_mcts_t *m, const void *head, size_t ld, int head_is_bf16, double c, uint32_t level_budget,
                      uint32_t max_states, rc_stream_t stream) {
    if (int rc = check_mcts(m)) return rc;
    RC_REQUIRE(head != nullptr, RC_ERR_NULL);
    RC_REQUIRE(ld >= (size_t)kActions + 1 && max_states > 0, RC_ERR_RANGE);
    const unsigned g = mcts_grid(m);
#define RC_STEP(NT_)                                                                                                                        \
    do {                                                                                                                                    \
        if (line_waves(g) == 4)                                                                                                              \
            hipLaunchKernelGGL((k_mcts_select<2, true, NT_, 4>), dim3(g), dim3(NT_), 0, (hipStream_t)stream, *m, c, level_budget, head,      \
                               (const float *)nullptr, ld, head_is_bf16 != 0, max_states);                                                  \
        else                                                                                                                                \
            hipLaunchKernelGGL((k_mcts_select<2, true, NT_, 1>), dim3(g), dim3(NT_), 0, (hipStream_t)stream, *m, c, level_budget, head,      \
                               (const float *)nullptr, ld, head_is_bf16 != 0, max_states);                                                  \
    } while (0)
    const unsigned nt = step_threads(g);
    if (nt == 1024) RC_STEP(1024);
    else if (nt == 512) RC_STEP(512);
    else RC_STEP(256);
#undef RC_STEP
    return launch_status();
}

}  // extern "C"
