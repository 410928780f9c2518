// gst_track.cpp -- which device ranges still hold the exact zeros of an analytic Jacobian.
//
// Tracked memory is OPT-IN per allocation: gst_device_malloc_tracked (the caller states that every write to it goes through
// this library or is announced with gst_device_touch); plain gst_device_malloc memory is the caller's and is never claimed.
// A claim remembers the plan that made it: its device word is only ever read and written on THAT plan's stream, so a row
// scaling issued through another plan (another stream: no ordering against the owner's next fill) drops the claim instead.
//
// Entry (element of circuit c, parameter of gate g) of an exact Jacobian is zero when c never applies g: 30 % of a GST
// Jacobian.  A repeated fill into the same destination need not store those zeros again (a third of the D = 16
// contraction's stores) -- provided nothing else was written there in between.  For memory the library handed out as TRACKED
// (gst_device_malloc_tracked) or owns (a plan's staging buffer) that is knowable: every entry point that writes device memory
// reports the range here (track_touch), a fill records what it left behind (track_claim_set), and the next fill with the
// same signature (plan, column request, leading dimension) finds the claim and skips the zeros.  Row scalings
// (gst_fill_jtj_dev) keep zeros zero unless a factor is not finite: the claim carries a device word that a check of the
// factors clears on the stream, and the contraction reads it.  Foreign pointers are never claimed (their owners may write
// them with their own kernels); GST_OPT_ANALYTIC_KEEP_ZEROS = 1 remains the caller's explicit promise for those.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <map>
#include <mutex>
#include <vector>
#include "gst_internal.hpp"

namespace gst {
namespace {

struct Claim { const char* base; size_t bytes; uint64_t sig; uint32_t* d_ok; int device; uint64_t owner; };

// (never destroyed: a plan released during process teardown may still report here)
std::mutex& g_mu = *new std::mutex;
std::map<const char*, size_t>& g_allocs = *new std::map<const char*, size_t>;              // tracked allocations: base -> bytes
std::vector<Claim>& g_claims = *new std::vector<Claim>;
std::vector<std::pair<int, uint32_t*>>& g_free_flags = *new std::vector<std::pair<int, uint32_t*>>;  // (device, word) of dropped claims, reused

bool overlaps(const char* a, size_t na, const char* b, size_t nb) { return a < b + nb && b < a + na; }

void drop_locked(size_t i)
{
    g_free_flags.push_back({g_claims[i].device, g_claims[i].d_ok});
    g_claims[i] = g_claims.back();
    g_claims.pop_back();
}

}  // namespace

void track_alloc(const void* p, size_t bytes)
{
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_mu);
    g_allocs[(const char*)p] = bytes;
}

void track_touch_locked(const char* p, size_t bytes)
{
    for (size_t i = 0; i < g_claims.size();)
        if (overlaps(g_claims[i].base, g_claims[i].bytes, p, bytes)) drop_locked(i); else i++;
}

void track_free(const void* p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_allocs.find((const char*)p);
    if (it == g_allocs.end()) return;
    track_touch_locked(it->first, it->second);
    g_allocs.erase(it);
}

bool track_owned(const void* p, size_t bytes)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_allocs.upper_bound((const char*)p);
    if (it == g_allocs.begin()) return false;
    --it;
    return (const char*)p >= it->first && (const char*)p + bytes <= it->first + it->second;
}

void track_touch(const void* p, size_t bytes)
{
    if (!p || !bytes) return;
    std::lock_guard<std::mutex> lk(g_mu);
    track_touch_locked((const char*)p, bytes);
}

uint32_t* track_claim_find(const void* base, size_t bytes, uint64_t sig)
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (const Claim& c : g_claims)
        if (c.base == (const char*)base && c.bytes == bytes && c.sig == sig) return c.d_ok;
    return nullptr;
}

uint32_t* track_claim_set(const void* base, size_t bytes, uint64_t sig, int device, uint64_t owner)
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (const Claim& c : g_claims)
        if (c.base == (const char*)base && c.bytes == bytes && c.sig == sig) return c.d_ok;      // (renewed by the caller's memset)
    track_touch_locked((const char*)base, bytes);
    uint32_t* w = nullptr;
    for (size_t i = 0; i < g_free_flags.size(); i++)
        if (g_free_flags[i].first == device) { w = g_free_flags[i].second; g_free_flags[i] = g_free_flags.back(); g_free_flags.pop_back(); break; }
    if (!w && hipMalloc((void**)&w, sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    g_claims.push_back({(const char*)base, bytes, sig, w, device, owner});
    return w;
}

uint32_t* track_claim_overlapping(const void* p, size_t bytes, bool* several, uint64_t* owner)
{
    std::lock_guard<std::mutex> lk(g_mu);
    uint32_t* w = nullptr;
    int n = 0;
    for (const Claim& c : g_claims)
        if (overlaps(c.base, c.bytes, (const char*)p, bytes)) { w = c.d_ok; n++; if (owner) *owner = c.owner; }
    if (several) *several = n > 1;
    return n == 1 ? w : nullptr;
}

}  // namespace gst
