// Batched weighted A* for MI355X (gfx950): B independent problems, each expanding its N best open
// nodes per iteration.  Restates the per-problem semantics of the reference's agent
// (librubiks/solving/agents.py:171-413) exactly: pop order by (cost, index), children in row order
// 12 p + k, "first occurrence" dedup inside the batch, new indices in row order, NumPy's
// gather-then-scatter semantics in the two relaxation passes.
//
// One 256-thread workgroup owns one problem per phase; its threads stride over the <= 12 N child
// rows.  Everything is integer / hashing work except the float64 cost; the value network runs
// outside (compacted over all problems) between the two kernels.
#include <limits.h>

#include "rubiks_common.h"

namespace rubiks {

constexpr int kA12 = kActions;
constexpr int kIdle = INT_MAX;

struct AStarView {   // per-problem slices of rc_astar_t
    uint4 *keys;
    int *G, *parents, *claim, *tab, *heap_idx, *popped, *child_node, *row_tmp;
    u8 *parent_actions, *row_flags;
    double *heap_cost;
    uint4 *child_keys;
    u32 mask;
};

__device__ __forceinline__ AStarView view_of(const rc_astar_t &a, u32 b) {
    AStarView v;
    const size_t base = (size_t)b * (a.capacity + 1);
    const size_t rows = (size_t)b * a.expansions * kA12;
    v.keys = reinterpret_cast<uint4 *>(a.keys) + base;
    v.G = a.G + base;
    v.parents = a.parents + base;
    v.parent_actions = a.parent_actions + base;
    v.claim = a.claim + base;
    v.tab = a.hash + (size_t)b * a.hash_size;
    v.heap_cost = a.heap_cost + base;
    v.heap_idx = a.heap_idx + base;
    v.popped = a.popped + (size_t)b * a.expansions;
    v.child_keys = reinterpret_cast<uint4 *>(a.child_keys) + rows;
    v.child_node = a.child_node + rows;
    v.row_tmp = a.row_tmp + rows;
    v.row_flags = a.row_flags + rows;
    v.mask = a.hash_size - 1;
    return v;
}

// ---- open list: binary min-heap ordered by (cost, index), operated by one lane ------------------
__device__ __forceinline__ bool heap_less(double ca, int ia, double cb, int ib) { return ca < cb || (ca == cb && ia < ib); }

__device__ void heap_push(double *hc, int *hi, int &size, double cost, int idx) {
    int i = size++;
    while (i > 0) {
        const int p = (i - 1) >> 1;
        const double pc = hc[p];
        const int pi = hi[p];
        if (!heap_less(cost, idx, pc, pi)) break;
        hc[i] = pc;
        hi[i] = pi;
        i = p;
    }
    hc[i] = cost;
    hi[i] = idx;
}

__device__ int heap_pop(double *hc, int *hi, int &size) {
    const int top = hi[0];
    --size;
    if (size > 0) {
        const double cost = hc[size];
        const int idx = hi[size];
        int i = 0;
        for (;;) {
            int c = 2 * i + 1;
            if (c >= size) break;
            double cc = hc[c];
            int ci = hi[c];
            if (c + 1 < size) {
                const double rc = hc[c + 1];
                const int ri = hi[c + 1];
                if (heap_less(rc, ri, cc, ci)) { ++c; cc = rc; ci = ri; }
            }
            if (!heap_less(cc, ci, cost, idx)) break;
            hc[i] = cc;
            hi[i] = ci;
            i = c;
        }
        hc[i] = cost;
        hi[i] = idx;
    }
    return top;
}

// ---- init ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_astar_init(rc_astar_t a, const u8 *__restrict__ roots, size_t stride) {
    const u32 b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= a.n_problems) return;
    AStarView v = view_of(a, b);
    u32 w[4] = {0, 0, 0, 0};
    bool solved = true;
#pragma unroll
    for (int j = 0; j < kPlanes; ++j) {
        const u32 code = roots[(size_t)j * stride + b] & 31u;
        key_set(w, j, code);
        solved &= code == (u32)(u8)kTables.solved[j];
    }
    const uint4 key = make_uint4(w[0], w[1], w[2], w[3]);
    v.keys[1] = key;
    v.tab[key_hash(key) & v.mask] = 1;
    v.G[1] = 0;
    v.parents[1] = 0;
    v.parent_actions[1] = 0;
    v.heap_cost[0] = 0.0;   // agents.py:234: cost 0, index 1
    v.heap_idx[0] = 1;
    a.heap_size[b] = solved ? 0 : 1;
    a.n_nodes[b] = solved ? 0 : 1;   // the reference returns before inserting a solved root (agents.py:230)
    a.status[b] = solved ? RC_ASTAR_ROOT_SOLVED : RC_ASTAR_RUNNING;
    a.solved_idx[b] = -1;
    a.iterations[b] = 0;
    a.n_popped[b] = 0;
    a.new_count[b] = 0;
}

// ---- pop + expand + dedup + append (agents.py:236-313) --------------------------------------------
__global__ __launch_bounds__(kBlock) void k_astar_pop_expand(rc_astar_t a, u32 max_states) {
    __shared__ u32 s_lut[sizeof(kTables.lut) / 4];
    __shared__ int s_scan[kBlock];
    __shared__ int s_npop, s_base;
    const u32 b = blockIdx.x, tid = threadIdx.x;
    stage_to_lds(s_lut, c_tables.lut, sizeof(kTables.lut));
    const u8 *lut = reinterpret_cast<const u8 *>(s_lut);
    AStarView v = view_of(a, b);
    if (tid == 0) {
        a.new_count[b] = 0;
        a.n_popped[b] = 0;
        int npop = 0;
        if (a.status[b] == RC_ASTAR_RUNNING) {
            const u32 n = (u32)a.n_nodes[b];
            if (n + a.expansions * kA12 > max_states || n + a.expansions * kA12 > a.capacity) {   // agents.py:236
                a.status[b] = RC_ASTAR_EXHAUSTED;
            } else {
                int size = a.heap_size[b];
                npop = min(size, (int)a.expansions);   // agents.py:238
                if (npop == 0) a.status[b] = RC_ASTAR_OPEN_EMPTY;
                for (int i = 0; i < npop; ++i) v.popped[i] = heap_pop(v.heap_cost, v.heap_idx, size);
                a.heap_size[b] = size;
                a.n_popped[b] = npop;
                if (npop) a.iterations[b] += 1;
            }
        }
        s_npop = npop;
    }
    __syncthreads();
    const int npop = s_npop;
    if (npop == 0) return;
    const int rows = npop * kA12;
    const int n_old = a.n_nodes[b];

    // pass 1: children (row 12 p + k = action k on popped parent p) and read-only membership test
    for (int r = tid; r < rows; r += kBlock) {
        const int p = v.popped[r / kA12];
        const u32 act = (u32)(r % kA12);
        const uint4 pk = v.keys[p];
        u32 w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < kPlanes; ++j)
            key_set(w, j, lut[act * (2 * kCodePad) + (j >= kCorners ? kCodePad : 0) + key_code(pk, j)]);
        const uint4 ck = make_uint4(w[0], w[1], w[2], w[3]);
        v.child_keys[r] = ck;
        u32 h = key_hash(ck) & v.mask;
        int found = 0;
        for (;;) {
            const int s = v.tab[h];
            if (s == 0) break;
            if (key_eq(v.keys[s], ck)) { found = s; break; }
            h = (h + 1) & v.mask;
        }
        v.child_node[r] = found ? found : -(int)h - 1;   // unseen: remember where the probe stopped
        if (found) atomicMin(&v.claim[found], r);        // first row that reaches an already known state
    }
    __syncthreads();
    // pass 2: unseen rows claim a slot; equal states elect their lowest row (np.unique first occurrence)
    for (int r = tid; r < rows; r += kBlock) {
        int cn = v.child_node[r];
        if (cn > 0) continue;
        const uint4 ck = v.child_keys[r];
        u32 h = (u32)(-cn - 1);
        for (;;) {
            int s = v.tab[h];
            if (s == 0) {
                s = atomicCAS(&v.tab[h], 0, -(r + 1));
                if (s == 0) break;
            }
            if (s < 0 && key_eq(v.child_keys[-s - 1], ck)) {
                atomicMax(&v.tab[h], -(r + 1));   // -(row+1): the larger value is the smaller row
                break;
            }
            h = (h + 1) & v.mask;   // occupied by another state (old node or another pending child)
        }
        v.child_node[r] = -(int)h - 1;
    }
    __syncthreads();
    // pass 3: flags, new indices by a block-wide exclusive scan in row order (agents.py:299-303)
    int local = 0;
    const int per = (rows + kBlock - 1) / kBlock;   // contiguous rows per thread keep row order in the scan
    const int r0 = tid * per, r1 = min(rows, r0 + per);
    for (int r = r0; r < r1; ++r) {
        const int cn = v.child_node[r];
        u8 f = 0;
        if (cn > 0) {
            if (v.claim[cn] == r) f = 2;
        } else if (v.tab[-cn - 1] == -(r + 1)) {
            f = 1;
            ++local;
        }
        v.row_flags[r] = f;
    }
    s_scan[tid] = local;
    __syncthreads();
    for (int o = 1; o < kBlock; o <<= 1) {   // Hillis-Steele inclusive scan
        const int x = (tid >= (u32)o) ? s_scan[tid - o] : 0;
        __syncthreads();
        s_scan[tid] += x;
        __syncthreads();
    }
    int rank = s_scan[tid] - local;
    if (tid == kBlock - 1) s_base = s_scan[tid];
    __syncthreads();
    for (int r = r0; r < r1; ++r) {
        const u8 f = v.row_flags[r];
        const int cn = v.child_node[r];
        if (f == 1) {
            const int idx = n_old + 1 + rank++;
            const int p = v.popped[r / kA12];
            v.tab[-cn - 1] = idx;
            v.keys[idx] = v.child_keys[r];
            v.G[idx] = v.G[p] + 1;                          // agents.py:311
            v.parent_actions[idx] = (u8)(r % kA12);         // agents.py:312
            v.parents[idx] = p;                             // agents.py:313
            v.child_node[r] = idx;
        } else if (f == 2) {
            v.claim[cn] = kIdle;                            // release the election scratch
        }
    }
    if (tid == 0) {
        a.new_count[b] = s_base;
        a.n_nodes[b] = n_old + s_base;
    }
}

// ---- compacted network input ------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_astar_gather_new(rc_astar_t a, const int *__restrict__ new_offset,
                                                           u8 *__restrict__ out, size_t stride) {
    const u32 b = blockIdx.x;
    const int cnt = a.new_count[b];
    if (cnt == 0) return;
    AStarView v = view_of(a, b);
    const int first = a.n_nodes[b] - cnt + 1;
    const size_t col0 = (size_t)new_offset[b];
    for (int i = threadIdx.x; i < cnt; i += kBlock) {
        const uint4 k = v.keys[first + i];
#pragma unroll
        for (int j = 0; j < kPlanes; ++j) out[(size_t)j * stride + col0 + i] = (u8)key_code(k, j);
    }
}

// ---- push + win check + relaxation (agents.py:315-328,333-367) ------------------------------------
__global__ __launch_bounds__(kBlock) void k_astar_push_relax(rc_astar_t a, const int *__restrict__ new_offset,
                                                           const float *__restrict__ values, double lambda) {
    __shared__ int s_won;
    const u32 b = blockIdx.x, tid = threadIdx.x;
    const int npop = a.n_popped[b];
    if (npop == 0) return;
    AStarView v = view_of(a, b);
    const int cnt = a.new_count[b];
    const int first = a.n_nodes[b] - cnt + 1;
    if (tid == 0) s_won = 0;
    __syncthreads();
    // win check on the new states only (agents.py:321); the lowest index is what indices[solved] finds
    u32 sw[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < kPlanes; ++j) key_set(sw, j, (u32)(u8)kTables.solved[j]);
    const uint4 solved_key = make_uint4(sw[0], sw[1], sw[2], sw[3]);
    for (int i = tid; i < cnt; i += kBlock)
        if (key_eq(v.keys[first + i], solved_key)) atomicMax(&s_won, first + i);
    // open-list pushes in index order (agents.py:315-317); one lane, the heap is a serial structure
    if (tid == 0) {
        int size = a.heap_size[b];
        const float *val = values + new_offset[b];
        for (int i = 0; i < cnt; ++i) {
            const int idx = first + i;
            const double cost = lambda * (double)v.G[idx] + (double)(-val[i]);   // agents.py:380-383
            heap_push(v.heap_cost, v.heap_idx, size, cost, idx);
        }
        a.heap_size[b] = size;
    }
    __syncthreads();
    if (s_won) {
        if (tid == 0) {
            a.status[b] = RC_ASTAR_SOLVED;
            a.solved_idx[b] = s_won;
        }
        return;   // the reference returns before relaxing (agents.py:322-323)
    }
    const int rows = npop * kA12;
    // relaxation, case 1: a shorter way to an already seen child (agents.py:354-359).
    // NumPy gathers every right-hand side before it scatters, hence read / barrier / write.
    for (int r = tid; r < rows; r += kBlock) {
        int t = 0;
        if (v.row_flags[r] == 2) {
            const int s = v.child_node[r], p = v.popped[r / kA12];
            const int g = v.G[p] + 1;
            if (g < v.G[s]) t = g;   // G >= 1 here, so 0 means "no update"
        }
        v.row_tmp[r] = t;
    }
    __syncthreads();
    for (int r = tid; r < rows; r += kBlock) {
        const int t = v.row_tmp[r];
        if (t) {
            const int s = v.child_node[r];
            v.G[s] = t;
            v.parent_actions[s] = (u8)(r % kA12);
            v.parents[s] = v.popped[r / kA12];
        }
    }
    __syncthreads();
    // case 2: the seen child is a shortcut to its expanded parent (agents.py:362-367); a parent hit
    // by several rows keeps the LAST one, as NumPy's fancy assignment does.
    for (int r = tid; r < rows; r += kBlock) {
        int t = 0;
        if (v.row_flags[r] == 2) {
            const int s = v.child_node[r], p = v.popped[r / kA12];
            const int g = v.G[s] + 1;
            if (g < v.G[p]) {
                t = g;
                atomicMin(&v.claim[p], -r);   // most negative = largest row
            }
        }
        v.row_tmp[r] = t;
    }
    __syncthreads();
    for (int r = tid; r < rows; r += kBlock) {
        const int t = v.row_tmp[r];
        if (t) {
            const int p = v.popped[r / kA12];
            if (v.claim[p] == -r) {
                v.G[p] = t;
                v.parent_actions[p] = (u8)((r % kA12) ^ 1);
                v.parents[p] = v.child_node[r];
            }
        }
    }
    __syncthreads();
    for (int r = tid; r < rows; r += kBlock)
        if (v.row_tmp[r]) v.claim[v.popped[r / kA12]] = kIdle;
}

}  // namespace rubiks

using namespace rubiks;

static int check_astar(const rc_astar_t *a) {
    RC_REQUIRE(a != nullptr, RC_ERR_NULL);
    RC_REQUIRE(a->keys && a->G && a->parents && a->parent_actions && a->claim && a->hash && a->heap_cost && a->heap_idx &&
                   a->heap_size && a->n_nodes && a->status && a->solved_idx && a->iterations && a->n_popped &&
                   a->new_count && a->popped && a->child_keys && a->child_node && a->row_tmp && a->row_flags,
               RC_ERR_NULL);
    RC_REQUIRE(a->n_problems > 0 && a->expansions > 0 && a->capacity >= 12 * a->expansions + 1, RC_ERR_RANGE);
    RC_REQUIRE((a->hash_size & (a->hash_size - 1)) == 0 && a->hash_size >= 2 * (a->capacity + 1), RC_ERR_RANGE);
    RC_REQUIRE(aligned16(a->keys) && aligned16(a->child_keys), RC_ERR_ALIGN);
    return RC_OK;
}

extern "C" {

int rc_astar_init(const rc_astar_t *a, const int8_t *roots_soa, size_t stride, rc_stream_t stream) {
    if (int rc = check_astar(a)) return rc;
    RC_CHECK_SOA(roots_soa, a->n_problems, stride);
    hipLaunchKernelGGL(k_astar_init, dim3(grid_for(a->n_problems, kBlock, 1 << 30)), dim3(kBlock), 0, (hipStream_t)stream,
                       *a, (const u8 *)roots_soa, stride);
    return launch_status();
}

int rc_astar_pop_expand(const rc_astar_t *a, uint32_t max_states, rc_stream_t stream) {
    if (int rc = check_astar(a)) return rc;
    hipLaunchKernelGGL(k_astar_pop_expand, dim3(a->n_problems), dim3(kBlock), 0, (hipStream_t)stream, *a, max_states);
    return launch_status();
}

int rc_astar_gather_new(const rc_astar_t *a, const int32_t *new_offset, int8_t *out_soa, size_t stride,
                        rc_stream_t stream) {
    if (int rc = check_astar(a)) return rc;
    RC_REQUIRE(new_offset && out_soa, RC_ERR_NULL);
    RC_REQUIRE(aligned16(out_soa) && (stride & 15u) == 0, RC_ERR_ALIGN);
    hipLaunchKernelGGL(k_astar_gather_new, dim3(a->n_problems), dim3(kBlock), 0, (hipStream_t)stream, *a, new_offset,
                       (u8 *)out_soa, stride);
    return launch_status();
}

int rc_astar_push_relax(const rc_astar_t *a, const int32_t *new_offset, const float *values, double lambda,
                        rc_stream_t stream) {
    if (int rc = check_astar(a)) return rc;
    RC_REQUIRE(new_offset && values, RC_ERR_NULL);
    hipLaunchKernelGGL(k_astar_push_relax, dim3(a->n_problems), dim3(kBlock), 0, (hipStream_t)stream, *a, new_offset,
                       values, lambda);
    return launch_status();
}

}  // extern "C"
