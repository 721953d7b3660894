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
//
// Address space is handed out in SIZE CLASSES (powers of two) and comes back: a released range is unmapped, the GPU's
// translations are flushed (forget_translations), and the range waits on its class's free list for the next reservation of that
// class -- a process that cycles through forests of many shapes holds at most one idle range per class and live overlap, not one
// per forest it ever built (rc_vmm_retired_bytes stops growing after the first cycle).
//
// Every reservation, map, release and reuse is recorded: the last kEvents in memory (rc_vmm_dump), all of them appended to the file
// RUBIKS_VMM_LOG names ("stderr" = the process's stderr), flushed line by line so that the record survives the abort that follows a
// GPU memory access fault; rc_vmm_classify / tools/vmm_classify.py say what a faulting address was at that moment (memory behind
// it / a reserved row without memory / a released range / never ours).
#include <unistd.h>

#include <cinttypes>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "rubiks_common.h"

using namespace rubiks;

namespace {

struct Range {
    size_t bytes = 0, chunk = 0;
    void *raw = nullptr;      // the reservation (a whole size class); the range handed out starts at the next multiple of `chunk`
    size_t raw_bytes = 0;
    int device = 0;
    uint32_t uses = 0;        // times this address range has been handed out (1 = fresh addresses)
    std::vector<hipMemGenericAllocationHandle_t> handles;   // one per chunk
    std::vector<char> mapped;
    size_t mapped_bytes = 0;
};
struct Idle {                 // a released reservation waiting on its class's free list: nothing mapped, translations flushed
    void *raw;
    size_t raw_bytes;
    int device;
    uint32_t uses;
};
struct Event {
    uint64_t seq;
    char op;                  // A the reservation behind the R / U that follows (base = its first address), R range handed out on fresh
                              // addresses, U on an idle reservation of the class, M map, X release (unmap), I reservation put on the idle
                              // list, F its address space freed instead, E a HIP call failed
    uintptr_t base;
    size_t a, b;              // A: raw bytes, 0;  R/U: bytes, chunk (rc = uses);  M: offset, bytes of the range that have memory behind them
                              // after the call (from offset on; chunk multiples);  X: mapped bytes given back, raw bytes;  I/F: raw bytes, uses;
                              // E: offset, bytes
    int rc;
};

std::mutex g_mu;
std::unordered_map<void *, Range> g_ranges;
std::multimap<std::pair<int, size_t>, Idle> g_idle;      // (device, raw bytes) -> idle reservations
size_t g_retired_bytes = 0;                               // address space on the idle lists
size_t g_quarantined_bytes = 0;                           // address space of ranges whose unmap / flush failed: never reused, never freed
constexpr size_t kRetireBudget = 32ull << 40;             // of the 128 TiB a process has: beyond it released ranges are freed
constexpr size_t kMinChunk = 2u << 20;
constexpr int kEvents = 512;
Event g_events[kEvents];
uint64_t g_seq = 0;
FILE *g_log = nullptr;
bool g_log_checked = false;

void record(char op, const void *base, size_t a, size_t b, int rc) {   // g_mu held
    Event &e = g_events[g_seq % kEvents];
    e = Event{g_seq, op, reinterpret_cast<uintptr_t>(base), a, b, rc};
    ++g_seq;
    if (!g_log_checked) {
        g_log_checked = true;
        const char *path = std::getenv("RUBIKS_VMM_LOG");
        if (path && *path) {
            // one file per process ("<path>.<pid>", RANK in front when a launcher set it): the ranks of one job and the child processes of
            // a test run inherit the variable, and their address spaces have nothing to do with each other
            std::string name = path;
            if (std::strcmp(path, "stderr") != 0) {
                if (const char *rank = getenv("RANK")) name += std::string(".rank") + rank;
                name += "." + std::to_string((long)getpid());
            }
            g_log = std::strcmp(path, "stderr") == 0 ? stderr : std::fopen(name.c_str(), "a");
        }
        if (g_log) std::fprintf(g_log, "# rubiks vmm log: seq op base a b rc  (A raw_bytes 0 | R/U bytes chunk uses | M offset covered | X mapped raw | I/F raw uses | E offset bytes)\n");
    }
    if (g_log) {
        std::fprintf(g_log, "%" PRIu64 " %c 0x%" PRIxPTR " %zu %zu %d\n", e.seq, e.op, e.base, e.a, e.b, e.rc);
        std::fflush(g_log);
    }
}

// One allocation per chunk, every chunk at an address that is a multiple of the chunk size.  Measured on MI355X / ROCm 7.2
// (profiles/r4_vmm_raw_probe.txt, r4_vmm_raw_probe2.txt): hipMemSetAccess returns hipErrorInvalidValue for a piece whose virtual
// address is not aligned to the piece's own size, and hipMemAddressReserve's alignment argument does not deliver that for more
// than 2 MiB -- so one chunk more than asked for is reserved and the range starts at the first multiple of the chunk size inside
// (pieces of 4 .. 64 MiB placed like that map and hold their data).  A map call costs ~10 us + ~15 us per 1 000 chunks the
// process has mapped already (80 000 chunks of 2 MiB: 47 s in all), and it returns only when the GPU has finished everything
// queued before it: large forests take larger chunks, and the host maps in few, large steps (MCTSForest.grow).
// A failed HIP call leaves its code as the thread's "last error", which the next hipGetLastError() of anybody -- torch checks it
// after every launch -- would report as its own: read it away.
int failed(hipError_t e, const void *base = nullptr, size_t a = 0, size_t b = 0) {
    (void)hipGetLastError();
    record('E', base, a, b, hip_rc(e));
    return hip_rc(e);
}

// hipMemUnmap does not make the GPU forget its translations of the unmapped addresses (ROCm 7.2 / MI355X, tools/vmm_remap_probe.hip,
// profiles/r4_vmm_remap_probe.txt, profiles/r5_vmm_remap_probe.txt): memory mapped LATER at such an address -- in the same reservation,
// or in a new one that was handed the freed addresses again -- is read and written through the stale translations by part of the
// chip (9 of 9 rounds with wrong data; 0 of 9 at addresses never mapped before, and 0 of 9 at the addresses of a hipFree'd
// block).  An ordinary hipMalloc + hipFree after the unmaps cures it (0 wrong rounds with it, in every run of the probe): hipFree's
// own unmapping is announced to the GPU, and that announcement covers everything.  rc_vmm_release does that before a range goes to
// the idle list, so by the time an address is mapped a second time no translation of its first life is left.
// Returns false when the flush could not be done (no memory for the block, a failing runtime call): the caller must then never
// hand the addresses out again.
bool forget_translations() {
    void *blk = nullptr;
    if (hipMalloc(&blk, 2u << 20) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    bool ok = hipMemset(blk, 0, 2u << 20) == hipSuccess;
    ok = (hipDeviceSynchronize() == hipSuccess) && ok;
    ok = (hipFree(blk) == hipSuccess) && ok;
    return ok;
}

hipMemAllocationProp device_prop(int device) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    return prop;
}

size_t size_class(size_t need) {   // the power of two >= need (>= 4 MiB)
    size_t c = 4u << 20;
    while (c < need) c <<= 1;
    return c;
}

void append(std::string &s, const char *fmt, ...) {
    char line[256];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(line, sizeof line, fmt, ap);
    va_end(ap);
    s += line;
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
    RC_REQUIRE(bytes > 0 && bytes <= (1ull << 46) && chunk_bytes >= kMinChunk && (chunk_bytes & (chunk_bytes - 1)) == 0 && chunk_bytes <= (1u << 30),
               RC_ERR_RANGE);
    Range r;
    r.chunk = chunk_bytes;
    r.bytes = (bytes + chunk_bytes - 1) / chunk_bytes * chunk_bytes;
    r.raw_bytes = size_class(r.bytes + (chunk_bytes > kMinChunk ? chunk_bytes : 0));
    std::lock_guard<std::mutex> lock(g_mu);
    if (hipError_t e = hipGetDevice(&r.device); e != hipSuccess) return failed(e);
    auto idle = g_idle.find({r.device, r.raw_bytes});
    if (idle != g_idle.end()) {            // an idle reservation of this class: its addresses again (flushed when it was released)
        r.raw = idle->second.raw;
        r.uses = idle->second.uses + 1;
        g_retired_bytes -= r.raw_bytes;
        g_idle.erase(idle);
    } else {
        if (hipError_t e = hipMemAddressReserve(&r.raw, r.raw_bytes, kMinChunk, nullptr, 0); e != hipSuccess) return failed(e, nullptr, r.raw_bytes, chunk_bytes);
        r.uses = 1;
    }
    void *base = reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(r.raw) + chunk_bytes - 1) / chunk_bytes * chunk_bytes);
    r.handles.assign(r.bytes / chunk_bytes, hipMemGenericAllocationHandle_t{});
    r.mapped.assign(r.bytes / chunk_bytes, 0);
    record('A', r.raw, r.raw_bytes, 0, RC_OK);
    record(r.uses == 1 ? 'R' : 'U', base, r.bytes, chunk_bytes, (int)r.uses);
    g_ranges.emplace(base, std::move(r));
    *out_base = base;
    return RC_OK;
}

// Backs [offset, offset + bytes) of the range with physical memory (chunks already mapped are left alone).  Host-synchronous;
// kernels running on other parts of the range are not disturbed.  *out_new_bytes: physical bytes this call added -- also when it
// fails part of the way (the chunks mapped until then stay mapped).
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
    size_t added = 0, done = c0;   // chunks c0 .. done - 1 have memory behind them when the loop ends
    int rc = RC_OK;
    for (size_t c = c0; c <= c1 && rc == RC_OK; ++c) {
        if (r.mapped[c]) {
            done = c + 1;
            continue;
        }
        const size_t run = r.chunk;
        hipMemGenericAllocationHandle_t h{};
        char *at = static_cast<char *>(base) + c * r.chunk;
        if (hipError_t err = hipMemCreate(&h, run, &prop, 0); err != hipSuccess) {
            rc = failed(err, base, c * r.chunk, run);
        } else if (hipError_t err = hipMemMap(at, run, 0, h, 0); err != hipSuccess) {
            (void)hipMemRelease(h);
            rc = failed(err, base, c * r.chunk, run);
        } else if (hipError_t err = hipMemSetAccess(at, run, &access, 1); err != hipSuccess) {
            (void)hipMemUnmap(at, run);
            (void)hipMemRelease(h);
            rc = failed(err, base, c * r.chunk, run);
        } else {
            r.handles[c] = h;
            r.mapped[c] = 1;
            r.mapped_bytes += run;
            added += run;
            done = c + 1;
        }
    }
    if (out_new_bytes) *out_new_bytes = added;
    if (added) record('M', base, c0 * r.chunk, (done - c0) * r.chunk, rc);
    return rc;
}

int rc_vmm_mapped_bytes(void *base, size_t *out_bytes) {
    RC_REQUIRE(base != nullptr && out_bytes != nullptr, RC_ERR_NULL);
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_ranges.find(base);
    RC_REQUIRE(it != g_ranges.end(), RC_ERR_RANGE);
    *out_bytes = it->second.mapped_bytes;
    return RC_OK;
}

// Per chunk of the range: 1 = memory behind it.  out_flags: [n_chunks] bytes, or NULL to ask for the count alone.
int rc_vmm_chunk_map(void *base, uint8_t *out_flags, size_t cap, size_t *out_chunks) {
    RC_REQUIRE(base != nullptr && out_chunks != nullptr, RC_ERR_NULL);
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_ranges.find(base);
    RC_REQUIRE(it != g_ranges.end(), RC_ERR_RANGE);
    const Range &r = it->second;
    *out_chunks = r.mapped.size();
    if (out_flags) {
        RC_REQUIRE(cap >= r.mapped.size(), RC_ERR_RANGE);
        std::memcpy(out_flags, r.mapped.data(), r.mapped.size());
    }
    return RC_OK;
}

int rc_vmm_retired_bytes(size_t *out_bytes) {
    RC_REQUIRE(out_bytes != nullptr, RC_ERR_NULL);
    std::lock_guard<std::mutex> lock(g_mu);
    *out_bytes = g_retired_bytes;
    return RC_OK;
}

// Unmaps and releases the memory.  The caller has synchronised with every kernel that uses the range.  The address range goes to
// the idle list of its size class (after forget_translations, so that nothing of its mappings is left in the GPU when the next
// reservation of the class maps memory there); only beyond kRetireBudget of idle address space is it freed instead.
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
        if (hipError_t err = hipMemUnmap(static_cast<char *>(base) + c * r.chunk, r.chunk); err != hipSuccess && rc == RC_OK) rc = failed(err, base, c * r.chunk, r.chunk);
        if (hipError_t err = hipMemRelease(r.handles[c]); err != hipSuccess && rc == RC_OK) rc = failed(err, base, c * r.chunk, r.chunk);
    }
    record('X', base, r.mapped_bytes, r.raw_bytes, rc);
    bool clean = rc == RC_OK;   // an unmap that failed leaves chunks mapped: such addresses are never handed out again
    if (r.mapped_bytes) {   // chunks are only ever unmapped here: a range without memory now never had any.  On the range's device, whichever is current
        int current = r.device;
        (void)hipGetDevice(&current);
        if (current != r.device) (void)hipSetDevice(r.device);
        if (!forget_translations()) {
            clean = false;
            record('E', base, 0, r.raw_bytes, RC_ERR_HIP_BASE);   // the flush failed: stale translations may remain
        }
        if (current != r.device) (void)hipSetDevice(current);
    }
    if (!clean) {
        // Neither on the idle list (a later mapping there could be read through a translation of this life) nor freed (the runtime
        // would hand the same addresses to the next reservation): the reservation is kept, unused, for the life of the process.
        g_quarantined_bytes += r.raw_bytes;
        record('Q', r.raw, r.raw_bytes, r.uses, rc);
    } else if (g_retired_bytes + r.raw_bytes <= kRetireBudget) {
        g_idle.insert({{r.device, r.raw_bytes}, Idle{r.raw, r.raw_bytes, r.device, r.uses}});
        g_retired_bytes += r.raw_bytes;
        record('I', r.raw, r.raw_bytes, r.uses, RC_OK);
    } else {
        hipError_t err = hipMemAddressFree(r.raw, r.raw_bytes);
        if (err != hipSuccess && rc == RC_OK) rc = failed(err, r.raw, 0, r.raw_bytes);
        record('F', r.raw, r.raw_bytes, r.uses, hip_rc(err));
    }
    g_ranges.erase(it);
    return rc;
}

// What `addr` is to this library right now.  *out_base / *out_offset: the range it lies in (its handed-out base; offset may be
// negative-as-size_t for the alignment slack in front of the base) or 0.
//   RC_VMM_ADDR_UNKNOWN   not inside any reservation of this library (live or idle)
//   RC_VMM_ADDR_MAPPED    a live range, the chunk has memory behind it
//   RC_VMM_ADDR_UNMAPPED  a live range, no memory behind the chunk: a row that was touched before it was mapped
//   RC_VMM_ADDR_SLACK     a live reservation, outside the range handed out (alignment slack / the rest of the size class)
//   RC_VMM_ADDR_IDLE      a released range waiting for reuse: nothing is mapped there
int rc_vmm_classify(const void *addr, int *out_kind, void **out_base, size_t *out_offset) {
    RC_REQUIRE(out_kind != nullptr, RC_ERR_NULL);
    const uintptr_t a = reinterpret_cast<uintptr_t>(addr);
    if (out_base) *out_base = nullptr;
    if (out_offset) *out_offset = 0;
    *out_kind = RC_VMM_ADDR_UNKNOWN;
    std::lock_guard<std::mutex> lock(g_mu);
    for (const auto &kv : g_ranges) {
        const Range &r = kv.second;
        const uintptr_t raw = reinterpret_cast<uintptr_t>(r.raw), base = reinterpret_cast<uintptr_t>(kv.first);
        if (a < raw || a >= raw + r.raw_bytes) continue;
        if (out_base) *out_base = kv.first;
        if (out_offset) *out_offset = a - base;
        *out_kind = (a < base || a >= base + r.bytes) ? RC_VMM_ADDR_SLACK : r.mapped[(a - base) / r.chunk] ? RC_VMM_ADDR_MAPPED : RC_VMM_ADDR_UNMAPPED;
        return RC_OK;
    }
    for (const auto &kv : g_idle) {
        const uintptr_t raw = reinterpret_cast<uintptr_t>(kv.second.raw);
        if (a < raw || a >= raw + kv.second.raw_bytes) continue;
        if (out_base) *out_base = kv.second.raw;
        if (out_offset) *out_offset = a - raw;
        *out_kind = RC_VMM_ADDR_IDLE;
        return RC_OK;
    }
    return RC_OK;
}

// Text description of the node store's state: live ranges (with their mapped chunk runs), idle ranges per class, the last events.
// Writes at most cap - 1 characters + NUL into out (may be NULL); *out_needed = the full length incl. NUL.
int rc_vmm_dump(char *out, size_t cap, size_t *out_needed) {
    std::string s;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        append(s, "rubiks vmm: %zu live ranges, %zu idle (%zu bytes of address space), %" PRIu64 " events\n", g_ranges.size(), g_idle.size(),
               g_retired_bytes, g_seq);
        for (const auto &kv : g_ranges) {
            const Range &r = kv.second;
            append(s, "live base=0x%" PRIxPTR " bytes=%zu chunk=%zu raw=0x%" PRIxPTR " raw_bytes=%zu device=%d uses=%u mapped_bytes=%zu chunks:",
                   reinterpret_cast<uintptr_t>(kv.first), r.bytes, r.chunk, reinterpret_cast<uintptr_t>(r.raw), r.raw_bytes, r.device, r.uses, r.mapped_bytes);
            size_t runs = 0;
            for (size_t c = 0; c < r.mapped.size();) {
                if (!r.mapped[c]) {
                    ++c;
                    continue;
                }
                size_t e = c;
                while (e < r.mapped.size() && r.mapped[e]) ++e;
                if (++runs <= 64) append(s, " %zu-%zu", c, e - 1);
                c = e;
            }
            append(s, runs > 64 ? " ... (%zu runs)\n" : "\n", runs);
        }
        for (const auto &kv : g_idle)
            append(s, "idle raw=0x%" PRIxPTR " raw_bytes=%zu device=%d uses=%u\n", reinterpret_cast<uintptr_t>(kv.second.raw), kv.second.raw_bytes,
                   kv.second.device, kv.second.uses);
        const uint64_t first = g_seq > (uint64_t)kEvents ? g_seq - kEvents : 0;
        for (uint64_t q = first; q < g_seq; ++q) {
            const Event &e = g_events[q % kEvents];
            append(s, "event %" PRIu64 " %c 0x%" PRIxPTR " %zu %zu %d\n", e.seq, e.op, e.base, e.a, e.b, e.rc);
        }
    }
    if (out_needed) *out_needed = s.size() + 1;
    if (out && cap) {
        const size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
        std::memcpy(out, s.data(), n);
        out[n] = 0;
    }
    return RC_OK;
}

}  // extern "C"
