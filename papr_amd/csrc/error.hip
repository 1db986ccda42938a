#include "papr_common.h"
#include <stdarg.h>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

void papr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* papr_last_error(void) { return g_err; }
extern "C" int papr_abi_version(void) { return 27; }

// ---- process-wide switches and per-device caches ------------------------------------------------
#include <atomic>
namespace {
std::atomic<int32_t> g_switch[PAPR_SW_COUNT] = {{0}, {1}, {3}, {1}, {0}, {600}, {0}, {0}, {1}, {1}, {6130}, {0}, {1}, {1}};        // defaults = the product (papr_hip.h: PAPR_SW_*)
constexpr int MAX_DEVICES = 64;
std::atomic<int> g_cu[MAX_DEVICES];
std::atomic<unsigned> g_once[MAX_DEVICES];
int current_device() { int dev = 0; (void)hipGetDevice(&dev); return dev >= 0 && dev < MAX_DEVICES ? dev : 0; }
}  // namespace

int papr_switch(int which) { return which >= 0 && which < PAPR_SW_COUNT ? g_switch[which].load(std::memory_order_relaxed) : 0; }

extern "C" int papr_set_switch(int32_t which, int32_t value) {
    PAPR_REQUIRE(which >= 0 && which < PAPR_SW_COUNT, "papr_set_switch: unknown switch %d", which);
    g_switch[which].store(value, std::memory_order_relaxed);
    return 0;
}

extern "C" int32_t papr_get_switch(int32_t which) { return papr_switch(which); }

int papr_cu_count() {
    const int dev = current_device();
    int n = g_cu[dev].load(std::memory_order_relaxed);
    if (n <= 0) {
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (n <= 0) n = 256;
        g_cu[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

bool papr_first_on_device(int slot) {
    const unsigned bit = 1u << slot;
    return !(g_once[current_device()].fetch_or(bit, std::memory_order_relaxed) & bit);
}

// ---- optional launch timing (diagnostics only; see papr_profile_enable in papr_hip.h) ----------
namespace {
struct Pending { papr_profile_record rec; hipEvent_t a, b; };
std::mutex g_mu;
std::vector<Pending> g_log;
bool g_on = false;
}  // namespace

bool papr_prof_on() { return g_on; }

void papr_prof_begin(int kernel, long M, int N, int K, hipStream_t s) { papr_prof_begin2(kernel, M, N, K, 0, 0, s); }

void papr_prof_begin2(int kernel, long M, int N, int K, long long bytes, long long flops, hipStream_t s) {
    Pending p;
    p.rec.kernel = kernel; p.rec.M = M; p.rec.N = N; p.rec.K = K; p.rec.ms = 0.f; p.rec.bytes = bytes; p.rec.flops = flops;
    (void)hipEventCreate(&p.a);
    (void)hipEventCreate(&p.b);
    (void)hipEventRecord(p.a, s);
    std::lock_guard<std::mutex> lk(g_mu);
    g_log.push_back(p);
}

void papr_prof_end(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_log.empty()) (void)hipEventRecord(g_log.back().b, s);
}

extern "C" int papr_profile_enable(int on) {
    g_on = on != 0;
    return 0;
}

extern "C" int papr_profile_collect(papr_profile_record* out, int cap) {
    std::lock_guard<std::mutex> lk(g_mu);
    int n = (int)g_log.size();
    for (int i = 0; i < n; ++i) {
        Pending& p = g_log[i];
        (void)hipEventSynchronize(p.b);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, p.a, p.b);
        p.rec.ms = ms;
        if (out && i < cap) out[i] = p.rec;
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    g_log.clear();
    return n;
}
