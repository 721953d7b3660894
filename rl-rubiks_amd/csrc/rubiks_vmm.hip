// Node storage on demand (HIP virtual memory management).
//
// The reference grows a tree's node arrays by doubling whenever they fill (librubiks/solving/agents.py:450-459), so a search
// with max_states = 175 000 only ever pays for the nodes it creates.  The forests here are tree-major arrays with
// capacity + 1 rows per tree: allocated up front, 1 024 trees x 175 000 nodes x 285 B are 51 GB although a depth-20 tree ends at
// 12-14 k nodes on average, and 8 192 trees at that cap (408 GB) do not fit the 288 GB of HBM at all.
// Here an array is a RESERVED virtual address range of its full size; physical HBM is created and mapped into it chunk by
// chunk (a multiple of the allocation granularity, 2 MiB on MI355X) as the trees grow.  Addresses never change, so the kernels,
// the rc_mcts_t struct and every captured HIP graph stay as they are; the host maps ahead of the trees' growth at the points
// where it already looks at their node counts (MCTSRun.round).
#include <mutex>
#include <unordered_map>
#include <vector>

#include "rubiks_common.h"

using namespace rubiks;

namespace {

struct Range {
    size_t bytes = 0, chunk = 0;
    void *raw = nullptr;      // what hipMemAddressReserve returned (the range handed out starts at the next multiple of `chunk`)
    size_t raw_bytes = 0;
    int device = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;   // one per chunk
    std::vector<char> mapped;
    size_t mapped_bytes = 0;
    bool ever_mapped = false;
};
std::mutex g_mu;
std::unordered_map<void *, Range> g_ranges;
size_t g_retired_bytes = 0;                    // address space of released ranges that stays reserved (see rc_vmm_release)
constexpr size_t kRetireBudget = 32ull << 40;   // of the 128 TiB a process has

// One allocation per chunk, every chunk at an address that is a multiple of the chunk size.  Measured on MI355X / ROCm 7.2
// (profiles/r4_vmm_raw_probe.txt, r4_vmm_raw_probe2.txt): hipMemSetAccess returns hipErrorInvalidValue for a piece whose virtual
// address is not aligned to the piece's own size, and hipMemAddressReserve's alignment argument does not deliver that for more
// than 2 MiB -- so one chunk more than asked for is reserved and the range starts at the first multiple of the chunk size inside
// (pieces of 4 .. 64 MiB placed like that map and hold their data).  A map call costs ~10 us + ~15 us per 1 000 chunks the
// process has mapped already (80 000 chunks of 2 MiB: 47 s in all), and it returns only when the GPU has finished everything
// queued before it: large forests take larger chunks, and the host maps in few, large steps (MCTSForest.grow).
// A failed HIP call leaves its code as the thread's "last error", which the next hipGetLastError() of anybody -- torch checks it
// after every launch -- would report as its own: read it away.
int failed(hipError_t e) {
    (void)hipGetLastError();
    return hip_rc(e);
}

// hipMemUnmap does not make the GPU forget its translations of the unmapped addresses (ROCm 7.2 / MI355X, tools/vmm_remap_probe.hip,
// profiles/r4_vmm_remap_probe.txt): memory mapped LATER at such an address -- in the same reservation, or in a new one that was
// handed the freed addresses again -- is read and written through the stale translations by part of the chip (9 of 9 rounds with
// wrong data; 0 of 9 at addresses never mapped before, and 0 of 9 at the addresses of a hipFree'd block).  An ordinary
// hipMalloc + hipFree after the unmaps cures it in the probe (0 of 9): hipFree's own unmapping is announced to the GPU, and that
// announcement covers everything.  rc_vmm_release does that AND keeps the range's addresses reserved for good, so that neither a
// later reservation nor hipMalloc can be handed them: either measure alone passes the probe, address space is not scarce.
void forget_translations() {
    void *blk = nullptr;
    if (hipMalloc(&blk, 2u << 20) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    (void)hipMemset(blk, 0, 2u << 20);
    (void)hipDeviceSynchronize();
    (void)hipFree(blk);
}

hipMemAllocationProp device_prop(int device) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    return prop;
}

}  // namespace

extern "C" {

int rc_vmm_granularity(size_t *out_bytes) {
    RC_REQUIRE(out_bytes != nullptr, RC_ERR_NULL);
    int device = 0;
    if (hipError_t e = hipGetDevice(&device); e != hipSuccess) return hip_rc(e);
    const hipMemAllocationProp prop = device_prop(device);
    return hip_rc(hipMemGetAllocationGranularity(out_bytes, &prop, hipMemAllocationGranularityRecommended));
}

int rc_vmm_reserve(size_t bytes, size_t chunk_bytes, void **out_base) {
    RC_REQUIRE(out_base != nullptr, RC_ERR_NULL);
    *out_base = nullptr;
    constexpr size_t kMin = 2u << 20;
    RC_REQUIRE(bytes > 0 && chunk_bytes >= kMin && (chunk_bytes & (chunk_bytes - 1)) == 0 && chunk_bytes <= (1u << 30), RC_ERR_RANGE);
    Range r;
    r.chunk = chunk_bytes;
    r.bytes = (bytes + chunk_bytes - 1) / chunk_bytes * chunk_bytes;
    r.raw_bytes = r.bytes + (chunk_bytes > kMin ? chunk_bytes : 0);
    if (hipError_t e = hipGetDevice(&r.device); e != hipSuccess) return failed(e);
    if (hipError_t e = hipMemAddressReserve(&r.raw, r.raw_bytes, kMin, nullptr, 0); e != hipSuccess) return failed(e);
    void *base = reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(r.raw) + chunk_bytes - 1) / chunk_bytes * chunk_bytes);
    r.handles.assign(r.bytes / chunk_bytes, hipMemGenericAllocationHandle_t{});
    r.mapped.assign(r.bytes / chunk_bytes, 0);
    std::lock_guard<std::mutex> lock(g_mu);
    g_ranges.emplace(base, std::move(r));
    *out_base = base;
    return RC_OK;
}

// Backs [offset, offset + bytes) of the range with physical memory (chunks already mapped are left alone).  Host-synchronous;
// kernels running on other parts of the range are not disturbed.  *out_new_bytes: physical bytes this call added.
int rc_vmm_map(void *base, size_t offset, size_t bytes, size_t *out_new_bytes) {
    if (out_new_bytes) *out_new_bytes = 0;
    RC_REQUIRE(base != nullptr, RC_ERR_NULL);
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_ranges.find(base);
    RC_REQUIRE(it != g_ranges.end(), RC_ERR_RANGE);
    Range &r = it->second;
    RC_REQUIRE(offset <= r.bytes && bytes <= r.bytes - offset, RC_ERR_RANGE);
    if (bytes == 0) return RC_OK;
    const size_t c0 = offset / r.chunk, c1 = (offset + bytes - 1) / r.chunk;
    const hipMemAllocationProp prop = device_prop(r.device);
    hipMemAccessDesc access = {};
    access.location = prop.location;
    access.flags = hipMemAccessFlagsProtReadWrite;
    for (size_t c = c0; c <= c1; ++c) {
        if (r.mapped[c]) continue;
        const size_t run = r.chunk;
        hipMemGenericAllocationHandle_t h{};
        if (hipError_t err = hipMemCreate(&h, run, &prop, 0); err != hipSuccess) return failed(err);
        char *at = static_cast<char *>(base) + c * r.chunk;
        if (hipError_t err = hipMemMap(at, run, 0, h, 0); err != hipSuccess) {
            (void)hipMemRelease(h);
            return failed(err);
        }
        if (hipError_t err = hipMemSetAccess(at, run, &access, 1); err != hipSuccess) {
            (void)hipMemUnmap(at, run);
            (void)hipMemRelease(h);
            return failed(err);
        }
        r.handles[c] = h;
        r.mapped[c] = 1;
        r.ever_mapped = true;
        r.mapped_bytes += run;
        if (out_new_bytes) *out_new_bytes += run;
    }
    return RC_OK;
}

int rc_vmm_mapped_bytes(void *base, size_t *out_bytes) {
    RC_REQUIRE(base != nullptr && out_bytes != nullptr, RC_ERR_NULL);
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_ranges.find(base);
    RC_REQUIRE(it != g_ranges.end(), RC_ERR_RANGE);
    *out_bytes = it->second.mapped_bytes;
    return RC_OK;
}

int rc_vmm_retired_bytes(size_t *out_bytes) {
    RC_REQUIRE(out_bytes != nullptr, RC_ERR_NULL);
    std::lock_guard<std::mutex> lock(g_mu);
    *out_bytes = g_retired_bytes;
    return RC_OK;
}

// Unmaps and releases the memory.  The caller has synchronised with every kernel that uses the range.  The ADDRESSES of a range
// that had memory mapped are retired, not freed (see forget_translations): nothing is ever mapped at them again.  Only once
// kRetireBudget of address space has been retired are ranges freed for reuse, relying on forget_translations alone.
int rc_vmm_release(void *base) {
    RC_REQUIRE(base != nullptr, RC_ERR_NULL);
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_ranges.find(base);
    RC_REQUIRE(it != g_ranges.end(), RC_ERR_RANGE);
    Range &r = it->second;
    int rc = RC_OK;
    const size_t n = r.mapped.size();
    for (size_t c = 0; c < n; ++c) {
        if (!r.mapped[c]) continue;
        if (hipError_t err = hipMemUnmap(static_cast<char *>(base) + c * r.chunk, r.chunk); err != hipSuccess && rc == RC_OK) rc = failed(err);
        if (hipError_t err = hipMemRelease(r.handles[c]); err != hipSuccess && rc == RC_OK) rc = failed(err);
    }
    if (r.ever_mapped) {   // on the range's device, whichever is current
        int current = r.device;
        (void)hipGetDevice(&current);
        if (current != r.device) (void)hipSetDevice(r.device);
        forget_translations();
        if (current != r.device) (void)hipSetDevice(current);
    }
    if (r.ever_mapped && g_retired_bytes + r.raw_bytes <= kRetireBudget) {
        g_retired_bytes += r.raw_bytes;
    } else if (hipError_t err = hipMemAddressFree(r.raw, r.raw_bytes); err != hipSuccess && rc == RC_OK) {
        rc = failed(err);
    }
    g_ranges.erase(it);
    return rc;
}

}  // extern "C"
