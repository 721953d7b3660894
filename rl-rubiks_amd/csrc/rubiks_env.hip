// Cube-environment kernels for MI355X (gfx950): multi_rotate, 12-child expansion, solved test,
// one-hot encoding, scrambling, and the AoS<->SoA boundary transposes.
//
// All of this is HBM-bound byte work (~1 table lookup per byte moved): the design rules are
// coalesced 16-byte accesses on the SoA planes, the 12-move tables staged in LDS, and enough
// workgroups (>> 256) to fill 8 XCDs.  No MFMA here by design.
#include "rubiks_common.h"

namespace rubiks {

// =================================================================================================
// multi_rotate: out[j][i] = lut[act[i]][kind(j)][in[j][i]]        (librubiks/cube/cube.py:256-263)
// Each lane owns 4*W consecutive cubes: one W-dword load per plane (W=4 -> 16 B/lane, 1 KiB/wave).
// =================================================================================================
// Streaming accesses of the read-dominated large-batch kernels (multi_rotate, is_solved) are non-temporal (`nt`):
// every byte is touched once, and keeping it out of the caches' replacement state is worth ~15 % of HBM bandwidth
// (5.15 -> 5.9 TB/s on multi_rotate, interleaved A/B in one process).  The write-dominated kernels (expand12, as_oh)
// measured 3-4 % SLOWER with nt stores in the same A/B, so they keep plain stores.
typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ uint4 ld16(const uint4 *p) {
    if (NT) {
        const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p));
        return make_uint4(t[0], t[1], t[2], t[3]);
    }
    return *p;
}
template <bool NT> __device__ __forceinline__ void st16(uint4 *p, const uint4 &v) {
    if (NT) {
        v4u t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<v4u *>(p));
    } else {
        *p = v;
    }
}
template <bool NT> __device__ __forceinline__ u32 ld4(const u32 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }

template <int W> struct DW;
template <> struct DW<1> { using T = u32; };
template <> struct DW<2> { using T = uint2; };
template <> struct DW<4> { using T = uint4; };

template <int W> __device__ __forceinline__ void unpack(const typename DW<W>::T &v, u32 (&w)[W]);
template <> __device__ __forceinline__ void unpack<1>(const u32 &v, u32 (&w)[1]) { w[0] = v; }
template <> __device__ __forceinline__ void unpack<2>(const uint2 &v, u32 (&w)[2]) { w[0] = v.x; w[1] = v.y; }
template <> __device__ __forceinline__ void unpack<4>(const uint4 &v, u32 (&w)[4]) { w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
template <int W> __device__ __forceinline__ typename DW<W>::T pack(const u32 (&w)[W]);
template <> __device__ __forceinline__ u32 pack<1>(const u32 (&w)[1]) { return w[0]; }
template <> __device__ __forceinline__ uint2 pack<2>(const u32 (&w)[2]) { return make_uint2(w[0], w[1]); }
template <> __device__ __forceinline__ uint4 pack<4>(const u32 (&w)[4]) { return make_uint4(w[0], w[1], w[2], w[3]); }

template <int W, int NT = 0>
__global__ __launch_bounds__(kBlock) void k_multi_rotate(const typename DW<W>::T *__restrict__ in,
                                                         const typename DW<W>::T *__restrict__ act,
                                                         typename DW<W>::T *__restrict__ out, size_t n_vec,
                                                         size_t sin_vec, size_t sout_vec) {
    __shared__ u32 s_lut[sizeof(kTables.lut) / 4];
    stage_to_lds(s_lut, c_tables.lut, sizeof(kTables.lut));
    __syncthreads();
    const u8 *lut = reinterpret_cast<const u8 *>(s_lut);

    for (size_t g = (size_t)blockIdx.x * kBlock + threadIdx.x; g < n_vec; g += (size_t)gridDim.x * kBlock) {
        u32 a[W];
        unpack<W>(act[g], a);
        u32 abase[4 * W];   // LDS byte offset of lut[action][0][0] for each of this lane's cubes
#pragma unroll
        for (int c = 0; c < 4 * W; ++c) abase[c] = ((a[c >> 2] >> (8 * (c & 3))) & (kActionPad - 1)) * (2 * kCodePad);
#pragma unroll
        for (int j = 0; j < kPlanes; ++j) {
            const int kofs = (j >= kCorners) ? kCodePad : 0;
            u32 v[W], r[W];
            if (NT & 2) {
                typedef unsigned nvec __attribute__((ext_vector_type(W)));
                const nvec t = __builtin_nontemporal_load(reinterpret_cast<const nvec *>(&in[(size_t)j * sin_vec + g]));
#pragma unroll
                for (int w = 0; w < W; ++w) v[w] = t[w];
            } else unpack<W>(in[(size_t)j * sin_vec + g], v);
#pragma unroll
            for (int w = 0; w < W; ++w) {
                u32 acc = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) acc |= (u32)lut[abase[4 * w + b] + kofs + code_of(v[w], b)] << (8 * b);
                r[w] = acc;
            }
            if (NT & 1) {
                typedef unsigned nvec __attribute__((ext_vector_type(W)));
                nvec t;
#pragma unroll
                for (int w = 0; w < W; ++w) t[w] = r[w];
                __builtin_nontemporal_store(t, reinterpret_cast<nvec *>(&out[(size_t)j * sout_vec + g]));
            } else out[(size_t)j * sout_vec + g] = pack<W>(r);
        }
    }
}

// =================================================================================================
// expand12: children[12p + k] = action k on parent p    (agents.py:277-281,513; train.py:285)
// A workgroup stages 4*BLOCK parents (one dword per lane per plane) in LDS, then produces the
// 48*BLOCK child bytes of every plane output-centrically: lane -> one 16-byte chunk of the child
// plane, whose 4 dwords are 4 consecutive children of ONE parent each (12 % 4 == 0), so a dword is
// a single 4-byte read of the action-minor table lut4[kind][code][4m..4m+3].  Stores are 16 B per
// lane, fully coalesced.
// =================================================================================================
__device__ __forceinline__ u32 zero_bytes_to_flags(u32 x) {
    // byte -> 1 if the byte of x is zero else 0 (exact for all byte values)
    const u32 nz = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;
    return (nz >> 7) ^ 0x01010101u;
}

// The twelve states one move from solved: child k of a parent is the solved cube iff the parent is kNearSolved[k] = the solved cube
// turned by rev(k) = k ^ 1 (cube.py:194-200).  Such a parent agrees with the solved cube on exactly 12 of its 20 cubies.
struct NearSolved {
    uint8_t code[kActions][kPlanes];
};
constexpr NearSolved make_near_solved() {
    NearSolved t{};
    for (int k = 0; k < kActions; ++k)
        for (int j = 0; j < kPlanes; ++j) t.code[k][j] = kTables.lut[k ^ 1][j >= kCorners ? 1 : 0][(int)kTables.solved[j]];
    return t;
}
static __constant__ NearSolved c_near_solved = make_near_solved();

// FLAGS: the launch also answers multi_is_solved for the parents and for all their children (one data-generation step of an ADI
// rollout asks for both, train.py:285-296): from the parents it has staged anyway, without reading a child.
template <int BLOCK, bool NT = false, bool FLAGS = false>
__global__ __launch_bounds__(BLOCK) void k_expand12(const u32 *__restrict__ par, uint4 *__restrict__ child,
                                                    size_t n_parents, size_t n_par_dw, size_t n_chunks,
                                                    size_t sp_dw, size_t sc_vec, u32 *__restrict__ parent_flags = nullptr,
                                                    uint4 *__restrict__ child_flags = nullptr) {
    constexpr int PB = 4 * BLOCK;   // parents per tile
    __shared__ u32 s_lut4[sizeof(kTables.lut4) / 4];
    __shared__ u32 s_stage[kPlanes * BLOCK];
    stage_to_lds(s_lut4, c_tables.lut4, sizeof(kTables.lut4));
    const u8 *lut4 = reinterpret_cast<const u8 *>(s_lut4);
    const u8 *stage = reinterpret_cast<const u8 *>(s_stage);
    const int tid = threadIdx.x;

    // Chunk q = tid + BLOCK*r (r = 0..2) of the tile covers child dwords 4q..4q+3; dword d holds
    // children of local parent d/3, actions 4*(d%3) .. 4*(d%3)+3.  Plane-invariant, so hoisted.
    u32 poff[3][4], moff[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32 d = 4u * (tid + BLOCK * r) + i;
            poff[r][i] = d / 3u;
            moff[r][i] = (d - 3u * poff[r][i]) * 4u;
        }

    // gridDim.y workgroups share a tile by child PLANES (few parents, e.g. an ADI rollout's 16 384: 64 tiles alone leave three quarters
    // of the chip idle): workgroup y emits planes j0 .. j1 - 1; the flags, which need all 20 planes of a parent, come from y = 0.
    const int ppg = kPlanes / (int)gridDim.y, j0 = (int)blockIdx.y * ppg, j1 = j0 + ppg;
    const bool flags_here = FLAGS && blockIdx.y == 0;
    const size_t n_tiles = ceil_div(n_parents, (size_t)PB);
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t dw0 = tile * BLOCK + tid;   // this lane's parent dword within a plane
        __syncthreads();                          // previous tile's readers are done (and lut4 staged)
#pragma unroll
        for (int j = 0; j < kPlanes; ++j)
            if (flags_here || (j >= j0 && j < j1)) s_stage[j * BLOCK + tid] = (dw0 < n_par_dw) ? ld4<NT>(&par[(size_t)j * sp_dw + dw0]) : 0u;
        __syncthreads();
        if (flags_here && dw0 < n_par_dw) {   // this lane's four parents: how many cubies sit solved (byte-wise count), then the rare full compare
            u32 cnt = 0;
#pragma unroll
            for (int j = 0; j < kPlanes; ++j)
                cnt += zero_bytes_to_flags((s_stage[j * BLOCK + tid] & 0x1f1f1f1fu) ^ ((u32)(u8)kTables.solved[j] * 0x01010101u));
            parent_flags[dw0] = zero_bytes_to_flags(cnt ^ 0x14141414u);                  // 20 of 20
            u32 cf[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                           // 48 flag bytes: children 12 p .. 12 p + 11 of the four parents
            const u32 twelve = zero_bytes_to_flags(cnt ^ 0x0c0c0c0cu);                   // candidates: exactly 12 cubies solved
            if (twelve) {
                for (int b = 0; b < 4; ++b) {
                    if (!((twelve >> (8 * b)) & 1u)) continue;
                    for (int k = 0; k < kActions; ++k) {
                        bool same = true;
                        for (int j = 0; j < kPlanes; ++j) same &= (stage[(j * BLOCK + tid) * 4 + b] & 31u) == c_near_solved.code[k][j];
                        if (same) cf[(12 * b + k) >> 2] |= 1u << (8 * ((12 * b + k) & 3));
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) st16<NT>(&child_flags[3 * dw0 + i], make_uint4(cf[4 * i], cf[4 * i + 1], cf[4 * i + 2], cf[4 * i + 3]));
        }
        const size_t q0 = tile * (3 * BLOCK);     // first child chunk of this tile
#pragma unroll
        for (int j = 0; j < kPlanes; ++j) {
            if (j < j0 || j >= j1) continue;
            const u32 kbase = (j >= kCorners) ? kCodePad * kActions : 0;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                u32 o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32 code = stage[j * PB + poff[r][i]] & 31u;
                    o[i] = *reinterpret_cast<const u32 *>(lut4 + kbase + code * kActions + moff[r][i]);
                }
                const size_t q = q0 + tid + BLOCK * r;
                if (q < n_chunks) st16<NT>(&child[(size_t)j * sc_vec + q], make_uint4(o[0], o[1], o[2], o[3]));
            }
        }
    }
}

// =================================================================================================
// is_solved: all 20 planes equal the solved code                       (librubiks/cube/cube.py:85-89)
// Lane owns 16 cubes (one 16-byte load per plane); the solved codes are compile-time immediates.
// =================================================================================================
__device__ __forceinline__ u32 flags_to_bits(u32 f) { return ((f * 0x01020408u) >> 24) & 0xfu; }

template <bool NT>
__global__ __launch_bounds__(kBlock) void k_is_solved(const uint4 *__restrict__ soa, uint4 *__restrict__ flags,
                                                      u16 *__restrict__ mask16, u32 *__restrict__ count, size_t n,
                                                      size_t n_vec, size_t stride_vec) {
    // The loop is wave-uniform (every lane of a wavefront makes the same trips) because the count
    // path below uses a ballot and cross-lane shuffles.
    for (size_t g0 = (size_t)blockIdx.x * kBlock; g0 < n_vec; g0 += (size_t)gridDim.x * kBlock) {
        const size_t g = g0 + threadIdx.x;
        const bool live = g < n_vec;
        u32 bits = 0;
        if (live) {
            uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < kPlanes; ++j) {
                const u32 s = 0x01010101u * (u32)(u8)kTables.solved[j];
                const uint4 v = ld16<NT>(&soa[(size_t)j * stride_vec + g]);
                acc.x |= v.x ^ s; acc.y |= v.y ^ s; acc.z |= v.z ^ s; acc.w |= v.w ^ s;
            }
            uint4 f = make_uint4(zero_bytes_to_flags(acc.x), zero_bytes_to_flags(acc.y), zero_bytes_to_flags(acc.z),
                                 zero_bytes_to_flags(acc.w));
            bits = flags_to_bits(f.x) | (flags_to_bits(f.y) << 4) | (flags_to_bits(f.z) << 8) | (flags_to_bits(f.w) << 12);
            const size_t first = g * 16;
            if (first + 16 > n) {   // ragged tail: padding cubes are never solved
                const u32 valid = (u32)(n - first);
                bits &= (1u << valid) - 1u;
                u32 ff[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    if ((u32)c >= valid) ff[c >> 2] &= ~(0xffu << (8 * (c & 3)));
                f = make_uint4(ff[0], ff[1], ff[2], ff[3]);
            }
            if (flags) flags[g] = f;
            if (mask16) mask16[g] = (u16)bits;
        }
        if (count) {
            // Solved cubes are rare: one wavefront ballot decides whether anybody needs the atomic at all.
            if (__ballot(bits != 0)) {
                u32 c = __popc(bits);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
                if ((threadIdx.x & (kWave - 1)) == 0) atomicAdd(count, c);
            }
        }
    }
}

// =================================================================================================
// as_oh: out[i][24 j + s[i][j]] = 1                                 (librubiks/cube/cube.py:265-277)
// Write-bound (1920 B f32 / 960 B bf16 per state).  A workgroup stages SB states in LDS, then every
// lane emits 16-byte chunks of the row-major output: chunk c of a row covers the one-hot positions
// of a single cubie (24 % 4 == 0, 24 % 8 == 0), so it needs exactly one staged byte.
// =================================================================================================
// (NT stores were measured here too: the write-only one-hot kernel loses 5-10 % with them, so it keeps plain stores)
template <int SB, bool BF16, bool NT = false>
__global__ __launch_bounds__(kBlock) void k_as_oh(const u32 *__restrict__ soa, uint4 *__restrict__ out, size_t n,
                                                  size_t n_dw, size_t stride_dw) {
    constexpr int CPR = BF16 ? 60 : 120;   // 16-byte chunks per row
    constexpr int EPC = BF16 ? 8 : 4;      // one-hot elements per chunk
    constexpr int CPJ = 24 / EPC;          // chunks per cubie
    __shared__ u32 s_stage[kPlanes * SB / 4];
    const u8 *stage = reinterpret_cast<const u8 *>(s_stage);
    const size_t row0 = (size_t)blockIdx.x * SB;
    for (int i = threadIdx.x; i < kPlanes * SB / 4; i += kBlock) {
        const int j = i / (SB / 4), w = i % (SB / 4);
        const size_t dw = row0 / 4 + w;
        s_stage[i] = (dw < n_dw) ? soa[(size_t)j * stride_dw + dw] : 0u;
    }
    __syncthreads();
    const size_t rows = (n - row0 < (size_t)SB) ? n - row0 : SB;
    uint4 *dst = out + row0 * CPR;
    const u32 total = (u32)rows * CPR;
    for (u32 x = threadIdx.x; x < total; x += kBlock) {
        const u32 r = x / CPR, c = x - r * CPR;
        const u32 j = c / CPJ, o = (c - j * CPJ) * EPC;
        const u32 s = stage[j * SB + r];
        const u32 rel = s - o;   // position of the 1 inside this chunk if < EPC
        uint4 v;
        if (BF16) {
            const u32 lo = 0x3f80u, hi = 0x3f800000u;   // bf16 1.0 in the low / high half
            v.x = rel == 0 ? lo : rel == 1 ? hi : 0u;
            v.y = rel == 2 ? lo : rel == 3 ? hi : 0u;
            v.z = rel == 4 ? lo : rel == 5 ? hi : 0u;
            v.w = rel == 6 ? lo : rel == 7 ? hi : 0u;
        } else {
            const u32 one = 0x3f800000u;
            v.x = rel == 0 ? one : 0u;
            v.y = rel == 1 ? one : 0u;
            v.z = rel == 2 ? one : 0u;
            v.w = rel == 3 ? one : 0u;
        }
        st16<NT>(&dst[x], v);
    }
}

// =================================================================================================
// apply_moves / sequence_states: scrambling                        (librubiks/cube/cube.py:206-234)
// A lane keeps 4 cubes (20 packed dwords) in registers and walks the move list.
// =================================================================================================
__device__ __forceinline__ void rotate_packed(u32 (&st)[kPlanes], u32 a4, const u8 *lut) {
    u32 abase[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) abase[b] = ((a4 >> (8 * b)) & (kActionPad - 1)) * (2 * kCodePad);
#pragma unroll
    for (int j = 0; j < kPlanes; ++j) {
        const int kofs = (j >= kCorners) ? kCodePad : 0;
        u32 acc = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) acc |= (u32)lut[abase[b] + kofs + code_of(st[j], b)] << (8 * b);
        st[j] = acc;
    }
}

__global__ __launch_bounds__(kBlock) void k_apply_moves(u32 *__restrict__ soa, const u32 *__restrict__ moves, size_t n_dw,
                                                        size_t stride_dw, size_t moves_stride_dw, size_t depth) {
    __shared__ u32 s_lut[sizeof(kTables.lut) / 4];
    stage_to_lds(s_lut, c_tables.lut, sizeof(kTables.lut));
    __syncthreads();
    const u8 *lut = reinterpret_cast<const u8 *>(s_lut);
    for (size_t g = (size_t)blockIdx.x * kBlock + threadIdx.x; g < n_dw; g += (size_t)gridDim.x * kBlock) {
        u32 st[kPlanes];
#pragma unroll
        for (int j = 0; j < kPlanes; ++j) st[j] = soa[(size_t)j * stride_dw + g];
        for (size_t d = 0; d < depth; ++d) rotate_packed(st, moves[d * moves_stride_dw + g], lut);
#pragma unroll
        for (int j = 0; j < kPlanes; ++j) soa[(size_t)j * stride_dw + g] = st[j];
    }
}

// One game per lane; the trajectory of a game is contiguous in the output (row g*depth + d).
__global__ __launch_bounds__(kBlock) void k_sequence_states(const u8 *__restrict__ moves, u8 *__restrict__ out, size_t games,
                                                            size_t depth, int with_solved, size_t stride) {
    __shared__ u32 s_lut[sizeof(kTables.lut) / 4];
    stage_to_lds(s_lut, c_tables.lut, sizeof(kTables.lut));
    __syncthreads();
    const u8 *lut = reinterpret_cast<const u8 *>(s_lut);
    for (size_t g = (size_t)blockIdx.x * kBlock + threadIdx.x; g < games; g += (size_t)gridDim.x * kBlock) {
        u32 st[kPlanes];
#pragma unroll
        for (int j = 0; j < kPlanes; ++j) st[j] = (u32)(u8)kTables.solved[j];
        size_t row = g * depth;
        if (with_solved) {
#pragma unroll
            for (int j = 0; j < kPlanes; ++j) out[(size_t)j * stride + row] = (u8)st[j];
            ++row;
        }
        const size_t steps = depth - (with_solved ? 1 : 0);
        for (size_t d = 0; d < steps; ++d, ++row) {
            rotate_packed(st, (u32)moves[d * games + g], lut);
#pragma unroll
            for (int j = 0; j < kPlanes; ++j) out[(size_t)j * stride + row] = (u8)st[j];
        }
    }
}

// =================================================================================================
// AoS (n,20) row-major  <->  SoA planes, 256 states per workgroup through LDS.
// =================================================================================================
constexpr int kTB = 256;   // states per transpose tile

__global__ __launch_bounds__(kBlock) void k_aos_to_soa(const u8 *__restrict__ aos, u32 *__restrict__ soa, size_t n,
                                                       size_t stride_dw) {
    __shared__ u32 s_tile[kTB * kPlanes / 4];
    u8 *tile = reinterpret_cast<u8 *>(s_tile);
    const size_t row0 = (size_t)blockIdx.x * kTB;
    const size_t rows = (n - row0 < (size_t)kTB) ? n - row0 : kTB;
    const u32 *src = reinterpret_cast<const u32 *>(aos + row0 * kPlanes);   // row0*20 is a multiple of 16
    const u32 n_src_dw = (u32)(rows * kPlanes + 3) / 4;
    const u32 full_dw = (u32)(rows * kPlanes) / 4;
    for (u32 i = threadIdx.x; i < kTB * kPlanes / 4; i += kBlock) {
        u32 v = 0;
        if (i < full_dw) v = src[i];
        else if (i < n_src_dw) {   // last partial dword of the array: byte loads stay inside the buffer
            const u8 *p = reinterpret_cast<const u8 *>(src + i);
            for (u32 b = 0; b < rows * kPlanes - 4 * full_dw; ++b) v |= (u32)p[b] << (8 * b);
        }
        s_tile[i] = v;
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < kPlanes * kTB / 4; i += kBlock) {
        const u32 j = i / (kTB / 4), w = i % (kTB / 4);
        if (4 * w >= round_up_dev16(rows)) continue;
        u32 v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) v |= (u32)tile[(4 * w + b) * kPlanes + j] << (8 * b);
        soa[(size_t)j * stride_dw + row0 / 4 + w] = v;
    }
}

__global__ __launch_bounds__(kBlock) void k_soa_to_aos(const u32 *__restrict__ soa, u8 *__restrict__ aos, size_t n,
                                                       size_t stride_dw) {
    __shared__ u32 s_tile[kPlanes * kTB / 4];
    const u8 *tile = reinterpret_cast<const u8 *>(s_tile);
    const size_t row0 = (size_t)blockIdx.x * kTB;
    const size_t rows = (n - row0 < (size_t)kTB) ? n - row0 : kTB;
    for (u32 i = threadIdx.x; i < kPlanes * kTB / 4; i += kBlock) {
        const u32 j = i / (kTB / 4), w = i % (kTB / 4);
        s_tile[i] = (4 * w < round_up_dev16(rows)) ? soa[(size_t)j * stride_dw + row0 / 4 + w] : 0u;
    }
    __syncthreads();
    u32 *dst = reinterpret_cast<u32 *>(aos + row0 * kPlanes);
    const u32 full_dw = (u32)(rows * kPlanes) / 4;
    for (u32 i = threadIdx.x; i < kTB * kPlanes / 4; i += kBlock) {
        u32 v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const u32 e = 4 * i + b;               // byte index inside the tile's AoS image
            const u32 r = e / kPlanes, j = e - r * kPlanes;
            v |= (u32)tile[j * kTB + r] << (8 * b);
        }
        if (i < full_dw) dst[i] = v;
        else {
            u8 *p = reinterpret_cast<u8 *>(dst + i);
            for (u32 b = 0; 4 * i + b < rows * kPlanes; ++b) p[b] = (u8)(v >> (8 * b));
        }
    }
}


// ---- small calls on the reference's own (n,20) row-major layout -----------------------------------------------
// The drop-in functions cube.rotate / multi_rotate / multi_is_solved / as_oh are called by the reference with
// n = 1 .. a few thousand (agents.py:109,513).  At those sizes the work is nothing and the cost is the number of
// launches and copies, so these kernels read and write the caller's row-major bytes directly -- the pointers may be
// pinned host memory mapped into the device (one launch, no staging copy, no transposition).
__global__ __launch_bounds__(kBlock) void k_multi_rotate_aos(const u8 *__restrict__ in, const u8 *__restrict__ actions,
                                                             u8 *__restrict__ out, size_t n) {
    __shared__ u32 s_lut[sizeof(kTables.lut) / 4];
    stage_to_lds(s_lut, c_tables.lut, sizeof(kTables.lut));
    __syncthreads();
    const u8 *lut = reinterpret_cast<const u8 *>(s_lut);
    for (size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x; idx < n * kPlanes; idx += (size_t)gridDim.x * kBlock) {
        const size_t i = idx / kPlanes;
        const u32 j = (u32)(idx - i * kPlanes);
        const u32 a = actions[i] & (kActionPad - 1);
        out[idx] = lut[a * (2 * kCodePad) + (j >= (u32)kCorners ? kCodePad : 0) + (in[idx] & 31u)];
    }
}

__global__ __launch_bounds__(kBlock) void k_is_solved_aos(const u8 *__restrict__ in, u8 *__restrict__ flags, size_t n) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < kPlanes; ++j) ok &= in[i * kPlanes + j] == (u8)kTables.solved[j];
        flags[i] = ok ? 1 : 0;
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_as_oh_aos(const u8 *__restrict__ in, T *__restrict__ out, size_t n, T one) {
    for (size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x; idx < n * (kPlanes * kCodes); idx += (size_t)gridDim.x * kBlock) {
        const size_t i = idx / (kPlanes * kCodes);
        const u32 c = (u32)(idx - i * (kPlanes * kCodes));
        out[idx] = in[i * kPlanes + c / kCodes] == (u8)(c % kCodes) ? one : (T)0;
    }
}

}  // namespace rubiks

// =================================================================================================
// C ABI
// =================================================================================================
using namespace rubiks;

template <bool BF16>
static int as_oh_impl(const int8_t *soa, void *out, size_t n, size_t stride, rc_stream_t stream) {
    if (n == 0) return RC_OK;
    RC_CHECK_SOA(soa, n, stride);
    RC_REQUIRE(out != nullptr, RC_ERR_NULL);
    RC_REQUIRE(aligned16(out), RC_ERR_ALIGN);
    hipStream_t s = (hipStream_t)stream;
    const size_t n_dw = round_up(n, 16) / 4;
    if (n >= ((size_t)1 << 16)) {
        constexpr int SB = 256;
        hipLaunchKernelGGL((k_as_oh<SB, BF16, false>), dim3((unsigned)ceil_div(n, SB)), dim3(kBlock), 0, s, (const u32 *)soa,
                           (uint4 *)out, n, n_dw, stride / 4);
    } else {
        constexpr int SB = 64;
        hipLaunchKernelGGL((k_as_oh<SB, BF16>), dim3((unsigned)ceil_div(n, SB)), dim3(kBlock), 0, s, (const u32 *)soa,
                           (uint4 *)out, n, n_dw, stride / 4);
    }
    return launch_status();
}

extern "C" {

int rc_multi_rotate(const int8_t *in_soa, const uint8_t *actions, int8_t *out_soa, size_t n, size_t stride_in,
                    size_t stride_out, rc_stream_t stream) {
    if (n == 0) return RC_OK;
    RC_CHECK_SOA(in_soa, n, stride_in);
    RC_CHECK_SOA(out_soa, n, stride_out);
    RC_REQUIRE(actions != nullptr, RC_ERR_NULL);
    RC_REQUIRE(aligned16(actions), RC_ERR_ALIGN);
    hipStream_t s = (hipStream_t)stream;
    // 16 cubes per lane once there are enough of them to fill the chip; 4 per lane below that.
    if (n >= ((size_t)1 << 20)) {
        const size_t n_vec = ceil_div(n, 16);
        // large batches: non-temporal loads and stores, one 16-cube vector per lane, no grid-stride cap
        hipLaunchKernelGGL((k_multi_rotate<4, 3>), dim3((unsigned)ceil_div(n_vec, (size_t)kBlock)), dim3(kBlock), 0, s,
                           (const uint4 *)in_soa, (const uint4 *)actions, (uint4 *)out_soa, n_vec, stride_in / 16,
                           stride_out / 16);
    } else {
        const size_t n_vec = ceil_div(n, 4);
        hipLaunchKernelGGL(k_multi_rotate<1>, dim3(grid_for(n_vec)), dim3(kBlock), 0, s, (const u32 *)in_soa,
                           (const u32 *)actions, (u32 *)out_soa, n_vec, stride_in / 4, stride_out / 4);
    }
    return launch_status();
}

int rc_expand12(const int8_t *parents_soa, int8_t *children_soa, size_t n_parents, size_t stride_p, size_t stride_c,
                rc_stream_t stream) {
    if (n_parents == 0) return RC_OK;
    RC_CHECK_SOA(parents_soa, n_parents, stride_p);
    RC_CHECK_SOA(children_soa, n_parents * kActions, stride_c);
    hipStream_t s = (hipStream_t)stream;
    const size_t n_par_dw = round_up(n_parents, 16) / 4;
    const size_t n_chunks = ceil_div(n_parents * kActions, 16);
    if (n_parents >= ((size_t)1 << 18)) {
        constexpr int BLOCK = 256;
        const size_t tiles = ceil_div(n_parents, 4 * BLOCK);
        hipLaunchKernelGGL((k_expand12<BLOCK, false>), dim3((unsigned)(tiles < 4096 ? tiles : 4096)), dim3(BLOCK), 0, s,
                           (const u32 *)parents_soa, (uint4 *)children_soa, n_parents, n_par_dw, n_chunks, stride_p / 4,
                           stride_c / 16);
    } else {
        constexpr int BLOCK = 64;
        const size_t tiles = ceil_div(n_parents, 4 * BLOCK);
        const unsigned split = tiles <= 128 ? 4u : tiles <= 512 ? 2u : 1u;
        hipLaunchKernelGGL(k_expand12<BLOCK>, dim3((unsigned)(tiles < 8192 ? tiles : 8192), split), dim3(BLOCK), 0, s,
                           (const u32 *)parents_soa, (uint4 *)children_soa, n_parents, n_par_dw, n_chunks, stride_p / 4,
                           stride_c / 16);
    }
    return launch_status();
}

int rc_expand12_flags(const int8_t *parents_soa, int8_t *children_soa, size_t n_parents, size_t stride_p, size_t stride_c,
                      uint8_t *parent_solved, uint8_t *child_solved, rc_stream_t stream) {
    if (n_parents == 0) return RC_OK;
    RC_CHECK_SOA(parents_soa, n_parents, stride_p);
    RC_CHECK_SOA(children_soa, n_parents * kActions, stride_c);
    RC_REQUIRE(parent_solved != nullptr && child_solved != nullptr, RC_ERR_NULL);
    RC_REQUIRE(aligned16(parent_solved) && aligned16(child_solved), RC_ERR_ALIGN);
    hipStream_t s = (hipStream_t)stream;
    const size_t n_par_dw = round_up(n_parents, 16) / 4;
    const size_t n_chunks = ceil_div(n_parents * kActions, 16);
    if (n_parents >= ((size_t)1 << 18)) {
        constexpr int BLOCK = 256;
        const size_t tiles = ceil_div(n_parents, 4 * BLOCK);
        hipLaunchKernelGGL((k_expand12<BLOCK, false, true>), dim3((unsigned)(tiles < 4096 ? tiles : 4096)), dim3(BLOCK), 0, s,
                           (const u32 *)parents_soa, (uint4 *)children_soa, n_parents, n_par_dw, n_chunks, stride_p / 4,
                           stride_c / 16, (u32 *)parent_solved, (uint4 *)child_solved);
    } else {
        constexpr int BLOCK = 64;
        const size_t tiles = ceil_div(n_parents, 4 * BLOCK);
        const unsigned split = tiles <= 128 ? 4u : tiles <= 512 ? 2u : 1u;   // planes shared by 4 / 2 workgroups while the tiles alone do not fill the chip
        hipLaunchKernelGGL((k_expand12<BLOCK, false, true>), dim3((unsigned)(tiles < 8192 ? tiles : 8192), split), dim3(BLOCK), 0, s,
                           (const u32 *)parents_soa, (uint4 *)children_soa, n_parents, n_par_dw, n_chunks, stride_p / 4,
                           stride_c / 16, (u32 *)parent_solved, (uint4 *)child_solved);
    }
    return launch_status();
}

int rc_is_solved(const int8_t *soa, uint8_t *flags, uint64_t *mask, uint32_t *count, size_t n, size_t stride,
                 rc_stream_t stream) {
    if (n == 0) return RC_OK;
    RC_CHECK_SOA(soa, n, stride);
    RC_REQUIRE(flags == nullptr || aligned16(flags), RC_ERR_ALIGN);
    RC_REQUIRE(mask == nullptr || aligned16(mask), RC_ERR_ALIGN);
    const size_t n_vec = ceil_div(n, 16);
    if (n >= ((size_t)1 << 20))
        hipLaunchKernelGGL(k_is_solved<true>, dim3((unsigned)ceil_div(n_vec, (size_t)kBlock)), dim3(kBlock), 0,
                           (hipStream_t)stream, (const uint4 *)soa, (uint4 *)flags, (u16 *)mask, count, n, n_vec, stride / 16);
    else
        hipLaunchKernelGGL(k_is_solved<false>, dim3(grid_for(n_vec)), dim3(kBlock), 0, (hipStream_t)stream,
                           (const uint4 *)soa, (uint4 *)flags, (u16 *)mask, count, n, n_vec, stride / 16);
    return launch_status();
}

int rc_as_oh_f32(const int8_t *soa, float *out, size_t n, size_t stride, rc_stream_t stream) {
    return as_oh_impl<false>(soa, out, n, stride, stream);
}
int rc_as_oh_bf16(const int8_t *soa, uint16_t *out, size_t n, size_t stride, rc_stream_t stream) {
    return as_oh_impl<true>(soa, out, n, stride, stream);
}

int rc_apply_moves(int8_t *soa, const uint8_t *moves, size_t n, size_t stride, size_t depth, rc_stream_t stream) {
    if (n == 0 || depth == 0) return RC_OK;
    RC_CHECK_SOA(soa, n, stride);
    RC_REQUIRE(moves != nullptr, RC_ERR_NULL);
    // moves is [depth][n] densely packed; dword access per 4 cubes needs n % 4 == 0 rows
    RC_REQUIRE((n & 3u) == 0 && ((uintptr_t)moves & 3u) == 0, RC_ERR_ALIGN);
    const size_t n_dw = n / 4;
    hipLaunchKernelGGL(k_apply_moves, dim3(grid_for(n_dw)), dim3(kBlock), 0, (hipStream_t)stream, (u32 *)soa,
                       (const u32 *)moves, n_dw, stride / 4, n / 4, depth);
    return launch_status();
}

int rc_sequence_states(const uint8_t *moves, int8_t *out_soa, size_t games, size_t depth, int with_solved,
                       size_t stride_out, rc_stream_t stream) {
    if (games == 0 || depth == 0) return RC_OK;
    RC_CHECK_SOA(out_soa, games * depth, stride_out);
    RC_REQUIRE(moves != nullptr, RC_ERR_NULL);
    hipLaunchKernelGGL(k_sequence_states, dim3(grid_for(games)), dim3(kBlock), 0, (hipStream_t)stream, moves,
                       (u8 *)out_soa, games, depth, with_solved, stride_out);
    return launch_status();
}

int rc_aos_to_soa(const int8_t *aos, int8_t *soa, size_t n, size_t stride, rc_stream_t stream) {
    if (n == 0) return RC_OK;
    RC_CHECK_SOA(soa, n, stride);
    RC_REQUIRE(aos != nullptr, RC_ERR_NULL);
    RC_REQUIRE(aligned16(aos), RC_ERR_ALIGN);
    hipLaunchKernelGGL(k_aos_to_soa, dim3((unsigned)ceil_div(n, kTB)), dim3(kBlock), 0, (hipStream_t)stream,
                       (const u8 *)aos, (u32 *)soa, n, stride / 4);
    return launch_status();
}

int rc_soa_to_aos(const int8_t *soa, int8_t *aos, size_t n, size_t stride, rc_stream_t stream) {
    if (n == 0) return RC_OK;
    RC_CHECK_SOA(soa, n, stride);
    RC_REQUIRE(aos != nullptr, RC_ERR_NULL);
    RC_REQUIRE(aligned16(aos), RC_ERR_ALIGN);
    hipLaunchKernelGGL(k_soa_to_aos, dim3((unsigned)ceil_div(n, kTB)), dim3(kBlock), 0, (hipStream_t)stream,
                       (const u32 *)soa, (u8 *)aos, n, stride / 4);
    return launch_status();
}

int rc_multi_rotate_aos(const int8_t *in_aos, const uint8_t *actions, int8_t *out_aos, size_t n, rc_stream_t stream) {
    if (n == 0) return RC_OK;
    RC_REQUIRE(in_aos && actions && out_aos, RC_ERR_NULL);
    hipLaunchKernelGGL(k_multi_rotate_aos, dim3(grid_for(n * kPlanes)), dim3(kBlock), 0, (hipStream_t)stream, (const u8 *)in_aos,
                       actions, (u8 *)out_aos, n);
    return launch_status();
}

int rc_is_solved_aos(const int8_t *in_aos, uint8_t *flags, size_t n, rc_stream_t stream) {
    if (n == 0) return RC_OK;
    RC_REQUIRE(in_aos && flags, RC_ERR_NULL);
    hipLaunchKernelGGL(k_is_solved_aos, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, (const u8 *)in_aos, flags, n);
    return launch_status();
}

int rc_as_oh_aos_f32(const int8_t *in_aos, float *out, size_t n, rc_stream_t stream) {
    if (n == 0) return RC_OK;
    RC_REQUIRE(in_aos && out, RC_ERR_NULL);
    hipLaunchKernelGGL(k_as_oh_aos<float>, dim3(grid_for(n * (kPlanes * kCodes))), dim3(kBlock), 0, (hipStream_t)stream, (const u8 *)in_aos, out,
                       n, 1.0f);
    return launch_status();
}

}  // extern "C"
