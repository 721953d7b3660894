// Breadth-first search over the cube graph for MI355X (gfx950), level-synchronous but with the exact
// bookkeeping of the reference's FIFO loop (librubiks/solving/agents.py:92-131):
//   * states are numbered in discovery order, so the frontier of a level is a contiguous node range and
//     child row 12 p + k (action k on the p-th frontier node) is the order the reference generates them;
//   * a row is a NEW state iff it is not in the table and it is the first row of the chunk with that
//     state (`if new_tstate in self.states: continue`, agents.py:109-110);
//   * the first solved row ends the search (agents.py:111-116); the solved state is never stored;
//   * `len(self) < max_states` is evaluated before each parent is popped (agents.py:105), so the first
//     parent whose preceding state count reaches max_states cuts the level.
// All of it is integer / hashing work: one thread per child row, 16-byte packed states, one 4-byte
// hash slot per probe.  The host reads back five words per chunk and decides commit / stop.
#include <hipcub/hipcub.hpp>

#include "rubiks_common.h"

namespace rubiks {

constexpr unsigned long long kNone = ~0ull;

constexpr uint4 make_solved_key() {
    u32 w[4] = {0, 0, 0, 0};
    for (int j = 0; j < kPlanes; ++j) w[j / 6] |= (u32)(u8)kTables.solved[j] << (5 * (j % 6));
    return uint4{w[0], w[1], w[2], w[3]};
}

__device__ __forceinline__ bool is_solved_key(const uint4 &k) {
    constexpr uint4 s = make_solved_key();
    return k.x == s.x && k.y == s.y && k.z == s.z && k.w == s.w;
}

// result words: [0] first solved row, [1] new rows in the chunk, [2] cutoff parent, [3] new rows before
// the solved row, [4] new rows before the cutoff parent
__global__ void k_bfs_init(rc_bfs_t b, const u8 *__restrict__ root) {
    u32 w[4] = {0, 0, 0, 0};
    for (int j = 0; j < kPlanes; ++j) key_set(w, j, root[j] & 31u);
    const uint4 key = make_uint4(w[0], w[1], w[2], w[3]);
    reinterpret_cast<uint4 *>(b.keys)[0] = key;
    b.parent[0] = 0;
    b.action[0] = 0;
    b.hash[key_hash(key) & (b.hash_size - 1)] = 1;
    b.result[0] = is_solved_key(key) ? 0ull : kNone;   // agents.py:100
}

__global__ void k_bfs_begin(rc_bfs_t b, u32 n_parents) {
    b.result[0] = kNone;
    b.result[1] = 0;
    b.result[2] = n_parents;
    b.result[3] = 0;
    b.result[4] = 0;
}

// children of frontier nodes lo .. lo + n_parents - 1, membership test against the committed table
__global__ __launch_bounds__(kBlock) void k_bfs_children(rc_bfs_t b, u32 lo, u32 rows) {
    __shared__ u32 s_lut[sizeof(kTables.lut) / 4];
    stage_to_lds(s_lut, c_tables.lut, sizeof(kTables.lut));
    __syncthreads();
    const u8 *lut = reinterpret_cast<const u8 *>(s_lut);
    const uint4 *keys = reinterpret_cast<const uint4 *>(b.keys);
    uint4 *child_keys = reinterpret_cast<uint4 *>(b.child_keys);
    const u32 mask = b.hash_size - 1;
    unsigned long long first_solved = kNone;
    for (u32 r = blockIdx.x * kBlock + threadIdx.x; r < rows; r += gridDim.x * kBlock) {
        const uint4 pk = keys[lo + r / kActions];
        const u32 act = r % kActions;
        u32 w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < kPlanes; ++j)
            key_set(w, j, lut[act * (2 * kCodePad) + (j >= kCorners ? kCodePad : 0) + key_code(pk, j)]);
        const uint4 ck = make_uint4(w[0], w[1], w[2], w[3]);
        child_keys[r] = ck;
        int slot_or_node;
        if (is_solved_key(ck)) {
            slot_or_node = 0;   // never stored (agents.py:111-116)
            if (first_solved == kNone) first_solved = r;
        } else {
            u32 h = key_hash(ck) & mask;
            int found = 0;
            for (;;) {
                const int s = b.hash[h];
                if (s == 0) break;
                if (key_eq(keys[s - 1], ck)) { found = s; break; }
                h = (h + 1) & mask;
            }
            slot_or_node = found ? found : -(int)h - 1;
        }
        b.child_slot[r] = slot_or_node;
    }
    if (first_solved != kNone) atomicMin(reinterpret_cast<unsigned long long *>(b.result), first_solved);
}

// unseen rows claim a slot; equal states elect their lowest row
__global__ __launch_bounds__(kBlock) void k_bfs_claim(rc_bfs_t b, u32 rows) {
    const uint4 *child_keys = reinterpret_cast<const uint4 *>(b.child_keys);
    const u32 mask = b.hash_size - 1;
    for (u32 r = blockIdx.x * kBlock + threadIdx.x; r < rows; r += gridDim.x * kBlock) {
        const int cs = b.child_slot[r];
        if (cs >= 0) continue;
        const uint4 ck = child_keys[r];
        u32 h = (u32)(-cs - 1);
        const int mine = -(int)(r + 1);
        for (;;) {
            int s = __hip_atomic_load(&b.hash[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (s == 0) {
                s = atomicCAS(&b.hash[h], 0, mine);
                if (s == 0) break;
            }
            if (s < 0 && key_eq(child_keys[-s - 1], ck)) {
                atomicMax(&b.hash[h], mine);   // -(row+1): the larger value is the smaller row
                break;
            }
            h = (h + 1) & mask;   // another state (committed node or another pending row)
        }
        b.child_slot[r] = -(int)h - 1;
    }
}

__global__ __launch_bounds__(kBlock) void k_bfs_flags(rc_bfs_t b, u32 rows) {
    for (u32 r = blockIdx.x * kBlock + threadIdx.x; r < rows; r += gridDim.x * kBlock) {
        const int cs = b.child_slot[r];
        b.flags[r] = (cs < 0 && __hip_atomic_load(&b.hash[-cs - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                                    -(int)(r + 1))
                         ? 1u
                         : 0u;
    }
}

// prefix = exclusive scan of flags.  One thread per parent finds the max_states cut (agents.py:105).
__global__ __launch_bounds__(kBlock) void k_bfs_cutoff(rc_bfs_t b, u32 n_parents, u32 n_nodes, u32 max_states) {
    const u32 rows = n_parents * kActions;
    for (u32 j = blockIdx.x * kBlock + threadIdx.x; j < n_parents; j += gridDim.x * kBlock) {
        const u32 before = n_nodes + b.prefix[j * kActions];
        const bool prev_ok = j == 0 || n_nodes + b.prefix[(j - 1) * kActions] < max_states;
        if (before >= max_states && prev_ok) {
            b.result[2] = j;
            b.result[4] = b.prefix[j * kActions];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        b.result[1] = b.prefix[rows - 1] + b.flags[rows - 1];
        const unsigned long long fs = b.result[0];
        if (fs != kNone) b.result[3] = b.prefix[fs];
    }
}

// appends the chunk's new states in row order and finalises their hash slots
__global__ __launch_bounds__(kBlock) void k_bfs_commit(rc_bfs_t b, u32 lo, u32 rows, u32 n_nodes) {
    uint4 *keys = reinterpret_cast<uint4 *>(b.keys);
    const uint4 *child_keys = reinterpret_cast<const uint4 *>(b.child_keys);
    for (u32 r = blockIdx.x * kBlock + threadIdx.x; r < rows; r += gridDim.x * kBlock) {
        if (!b.flags[r]) continue;
        const u32 idx = n_nodes + b.prefix[r];
        keys[idx] = child_keys[r];
        b.parent[idx] = lo + r / kActions;
        b.action[idx] = (u8)(r % kActions);
        b.hash[-b.child_slot[r] - 1] = (int)idx + 1;
    }
}

// actions from the root to `node`, written back to front (agents.py:112-115)
__global__ void k_bfs_path(rc_bfs_t b, u32 node, u8 *out, u32 *out_len, u32 max_len) {
    u32 n = 0;
    while (node != 0 && n < max_len) {
        out[n++] = b.action[node];
        node = b.parent[node];
    }
    *out_len = node == 0 ? n : ~0u;
}

static int check_bfs(const rc_bfs_t *b) {
    RC_REQUIRE(b != nullptr, RC_ERR_NULL);
    RC_REQUIRE(b->keys && b->parent && b->action && b->hash && b->child_keys && b->child_slot && b->flags && b->prefix &&
                   b->result && b->scan_tmp,
               RC_ERR_NULL);
    RC_REQUIRE(aligned16(b->keys) && aligned16(b->child_keys), RC_ERR_ALIGN);
    RC_REQUIRE(b->hash_size >= 2 && (b->hash_size & (b->hash_size - 1)) == 0, RC_ERR_RANGE);
    RC_REQUIRE(b->chunk >= 1 && (unsigned long long)b->chunk * kActions < (1ull << 31), RC_ERR_RANGE);
    RC_REQUIRE((unsigned long long)b->capacity * 2 <= b->hash_size, RC_ERR_RANGE);
    return RC_OK;
}

}  // namespace rubiks

using namespace rubiks;

extern "C" {

size_t rc_bfs_scan_bytes(uint32_t chunk) {
    size_t bytes = 0;
    const u32 *in = nullptr;
    u32 *out = nullptr;
    if (hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in, out, (int)(chunk * (size_t)kActions)) != hipSuccess) return 0;
    return round_up(bytes ? bytes : 16, 256);
}

int rc_bfs_init(const rc_bfs_t *b, const int8_t *root_state, rc_stream_t stream) {
    if (int rc = check_bfs(b)) return rc;
    RC_REQUIRE(root_state != nullptr, RC_ERR_NULL);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipError_t e = hipMemsetAsync(b->hash, 0, (size_t)b->hash_size * sizeof(int32_t), st); e != hipSuccess)
        return hip_rc(e);
    k_bfs_init<<<1, 1, 0, st>>>(*b, reinterpret_cast<const u8 *>(root_state));
    return launch_status();
}

int rc_bfs_expand(const rc_bfs_t *b, uint32_t lo, uint32_t n_parents, uint32_t n_nodes, uint32_t max_states,
                  rc_stream_t stream) {
    if (int rc = check_bfs(b)) return rc;
    RC_REQUIRE(n_parents >= 1 && n_parents <= b->chunk, RC_ERR_RANGE);
    RC_REQUIRE((unsigned long long)lo + n_parents <= n_nodes, RC_ERR_RANGE);
    RC_REQUIRE((unsigned long long)n_nodes + (unsigned long long)n_parents * kActions <= b->capacity, RC_ERR_RANGE);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const u32 rows = n_parents * kActions;
    const unsigned grid = grid_for(rows);
    k_bfs_begin<<<1, 1, 0, st>>>(*b, n_parents);
    k_bfs_children<<<grid, kBlock, 0, st>>>(*b, lo, rows);
    k_bfs_claim<<<grid, kBlock, 0, st>>>(*b, rows);
    k_bfs_flags<<<grid, kBlock, 0, st>>>(*b, rows);
    size_t bytes = b->scan_tmp_bytes;
    if (hipError_t e = hipcub::DeviceScan::ExclusiveSum(b->scan_tmp, bytes, b->flags, b->prefix, (int)rows, st);
        e != hipSuccess)
        return hip_rc(e);
    k_bfs_cutoff<<<grid_for(n_parents), kBlock, 0, st>>>(*b, n_parents, n_nodes, max_states);
    return launch_status();
}

int rc_bfs_commit(const rc_bfs_t *b, uint32_t lo, uint32_t n_parents, uint32_t n_nodes, rc_stream_t stream) {
    if (int rc = check_bfs(b)) return rc;
    RC_REQUIRE(n_parents >= 1 && n_parents <= b->chunk, RC_ERR_RANGE);
    RC_REQUIRE((unsigned long long)n_nodes + (unsigned long long)n_parents * kActions <= b->capacity, RC_ERR_RANGE);
    const u32 rows = n_parents * kActions;
    k_bfs_commit<<<grid_for(rows), kBlock, 0, static_cast<hipStream_t>(stream)>>>(*b, lo, rows, n_nodes);
    return launch_status();
}

int rc_bfs_path(const rc_bfs_t *b, uint32_t node, uint8_t *out_actions, uint32_t *out_len, uint32_t max_len,
                rc_stream_t stream) {
    if (int rc = check_bfs(b)) return rc;
    RC_REQUIRE(out_actions && out_len, RC_ERR_NULL);
    RC_REQUIRE(node < b->capacity, RC_ERR_RANGE);
    k_bfs_path<<<1, 1, 0, static_cast<hipStream_t>(stream)>>>(*b, node, out_actions, out_len, max_len);
    return launch_status();
}

}  // extern "C"
