// Is device memory mapped at a virtual address that was mapped BEFORE (to other physical memory) coherent?  Raw HIP, no library.
//     hipcc --offload-arch=gfx950 -O2 tools/vmm_remap_probe.hip -o build/vmm_remap_probe && build/vmm_remap_probe [rounds]
// Each round comes by address ranges in three ways, maps 2 MiB chunks into them, writes a pattern with one launch shape and
// checks it with two others (a kernel with another grid, and hipMemcpy to the host):
//   fresh      a range never handed out before (control; nothing of it was ever mapped)
//   in_place   the same reservation: hipMemUnmap + hipMemRelease of every chunk, then new chunks mapped at the same addresses
//   rereserve  hipMemUnmap + hipMemRelease + hipMemAddressFree, then hipMemAddressReserve again (the runtime tends to hand the
//              same addresses back) and map
//   hipfree    a hipMalloc'ed block is written and hipFree'd, then a range is reserved (does it land on the block's addresses?)
//              and mapped: what a reservation meets after torch.cuda.empty_cache()
// Prints one line per check that fails and a summary; exit code = number of failing (scenario, round) pairs.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                       \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) {                                                                     \
            std::fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            std::exit(100);                                                                         \
        }                                                                                           \
    } while (0)

constexpr size_t kChunk = 2u << 20;

__global__ void k_write(uint32_t *p, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i * 2654435761u + seed;
}

// another grid shape and a reversed walk: other CUs (and XCDs) read what k_write's wrote
__global__ void k_check(const uint32_t *p, size_t n, uint32_t seed, unsigned long long *bad) {
    unsigned long long mine = 0;
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
        const size_t i = n - 1 - j;
        mine += p[i] != (uint32_t)i * 2654435761u + seed;
    }
    if (mine) atomicAdd(bad, mine);
}

struct Range {
    void *base = nullptr;
    size_t bytes = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
};

static hipMemAllocationProp prop() {
    hipMemAllocationProp p = {};
    p.type = hipMemAllocationTypePinned;
    p.location.type = hipMemLocationTypeDevice;
    p.location.id = 0;
    return p;
}

static Range reserve(size_t chunks) {
    Range r;
    r.bytes = chunks * kChunk;
    CK(hipMemAddressReserve(&r.base, r.bytes, kChunk, nullptr, 0));
    return r;
}

static void map_all(Range &r) {
    const hipMemAllocationProp p = prop();
    hipMemAccessDesc acc = {};
    acc.location = p.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (size_t c = 0; c < r.bytes / kChunk; ++c) {
        hipMemGenericAllocationHandle_t h{};
        CK(hipMemCreate(&h, kChunk, &p, 0));
        CK(hipMemMap((char *)r.base + c * kChunk, kChunk, 0, h, 0));
        CK(hipMemSetAccess((char *)r.base + c * kChunk, kChunk, &acc, 1));
        r.handles.push_back(h);
    }
}

static void unmap_all(Range &r) {
    CK(hipDeviceSynchronize());
    for (size_t c = 0; c < r.handles.size(); ++c) {
        CK(hipMemUnmap((char *)r.base + c * kChunk, kChunk));
        CK(hipMemRelease(r.handles[c]));
    }
    r.handles.clear();
    if (std::getenv("PROBE_HIPFREE_AFTER_UNMAP")) {   // does an ordinary free (whose unmapping IS seen by the GPU) heal the ranges above?
        void *blk = nullptr;
        CK(hipMalloc(&blk, kChunk));
        CK(hipMemset(blk, 1, kChunk));
        CK(hipDeviceSynchronize());
        CK(hipFree(blk));
    }
}

static unsigned long long *g_bad = nullptr;
static std::vector<uint32_t> g_host;

// returns the number of failing checks (0..2)
static int exercise(const Range &r, const char *tag, int round, uint32_t seed) {
    const size_t n = r.bytes / 4;
    uint32_t *p = (uint32_t *)r.base;
    k_write<<<1024, 256>>>(p, n, seed);
    CK(hipMemset(g_bad, 0, 8));
    k_check<<<333, 192>>>(p, n, seed, g_bad);
    unsigned long long bad_k = 0;
    CK(hipMemcpy(&bad_k, g_bad, 8, hipMemcpyDeviceToHost));
    g_host.resize(n);
    CK(hipMemcpy(g_host.data(), p, r.bytes, hipMemcpyDeviceToHost));
    size_t bad_h = 0;
    for (size_t i = 0; i < n; ++i) bad_h += g_host[i] != (uint32_t)i * 2654435761u + seed;
    if (bad_k || bad_h)
        std::printf("  round %d %-9s %p +%zu MiB: kernel check %llu of %zu words wrong, host copy %zu wrong\n", round, tag, r.base, r.bytes >> 20, bad_k, n, bad_h);
    return (bad_k != 0) + (bad_h != 0);
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 8;
    CK(hipSetDevice(0));
    CK(hipMalloc(&g_bad, 8));
    int fails[4] = {0, 0, 0, 0};
    int landed = 0;
    std::vector<Range> retired;   // unmapped, never freed: their addresses cannot come back
    uint32_t seed = 1;
    for (int r = 0; r < rounds; ++r) {
        const size_t chunks = 8 + 4 * (r % 3);
        // fresh
        Range a = reserve(chunks);
        map_all(a);
        fails[0] += exercise(a, "fresh", r, seed++) != 0;
        // in_place
        unmap_all(a);
        map_all(a);
        fails[1] += exercise(a, "in_place", r, seed++) != 0;
        unmap_all(a);
        retired.push_back(a);
        // rereserve
        Range b = reserve(chunks);
        map_all(b);
        exercise(b, "fresh", r, seed++);
        unmap_all(b);
        const void *was = b.base;
        CK(hipMemAddressFree(b.base, b.bytes));
        Range c = reserve(chunks / 2), d = reserve(chunks / 2);
        map_all(c);
        map_all(d);
        const bool reused = (c.base >= was && (char *)c.base < (char *)was + b.bytes) || (d.base >= was && (char *)d.base < (char *)was + b.bytes);
        int f = exercise(c, "rereserve", r, seed++);
        f += exercise(d, "rereserve", r, seed++);
        fails[2] += f != 0;
        std::printf("round %d: freed range %p came back: %s\n", r, was, reused ? "yes" : "no");
        unmap_all(c);
        unmap_all(d);
        retired.push_back(c);
        retired.push_back(d);
        // hipfree
        void *blk = nullptr;
        CK(hipMalloc(&blk, chunks * kChunk));
        k_write<<<1024, 256>>>((uint32_t *)blk, chunks * kChunk / 4, seed++);
        CK(hipDeviceSynchronize());
        CK(hipFree(blk));
        Range e = reserve(chunks);
        map_all(e);
        const bool on_block = (char *)e.base < (char *)blk + chunks * kChunk && (char *)blk < (char *)e.base + e.bytes;
        landed += on_block;
        fails[3] += exercise(e, "hipfree", r, seed++) != 0;
        std::printf("round %d: reservation after hipFree(%p) at %p: %s the freed block\n", r, blk, e.base, on_block ? "ON" : "not on");
        unmap_all(e);
        retired.push_back(e);
    }
    std::printf("rounds with wrong data -- fresh: %d, in_place: %d, rereserve: %d, hipfree: %d (of %d each; %d hipfree reservations landed on the freed block)\n",
                fails[0], fails[1], fails[2], fails[3], rounds, landed);
    return fails[0] + fails[1] + fails[2] + fails[3];
}
