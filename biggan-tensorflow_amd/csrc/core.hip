// core.hip - ABI bookkeeping: version, thread-local error string, optional per-launch HIP-event
// timing of the MFMA GEMM families (bench.py's roofline leg).
#include <stdarg.h>

#include <mutex>
#include <utility>
#include <vector>

#include "common.h"

namespace bg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct ProfRec {
    hipEvent_t e0, e1;
    double flops, bytes;
    char tag[112];
    char kernel[48];
};
static thread_local char g_prof_kernel[48] = "";
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof_recs;     // completed-but-uncollected launches
static std::vector<ProfRec> g_prof_pool;     // reusable event pairs

bool prof_on() { return g_prof_on; }

void prof_kernel(const char* fmt, ...) {
    if (!g_prof_on) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_prof_kernel, sizeof(g_prof_kernel), fmt, ap);
    va_end(ap);
}

// (kernel, device) pairs that already carry the raised dynamic-LDS limit
static std::mutex g_lds_mu;
static std::vector<std::pair<const void*, int>> g_lds_done;

bool lds_opt_in(const void* fn, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lk(g_lds_mu);
    for (auto& e : g_lds_done)
        if (e.first == fn && e.second == dev) return true;
    if (bytes > 49152 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
        set_error("cannot raise the dynamic LDS limit of a kernel to %d bytes on device %d", bytes, dev);
        return false;
    }
    g_lds_done.emplace_back(fn, dev);
    return true;
}

ProfScope::ProfScope(hipStream_t s, double flops, const char* tag, double bytes) : stream(s), slot(-1) {
    if (!g_prof_on) return;
    g_prof_kernel[0] = 0;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r;
    if (!g_prof_pool.empty()) {
        r = g_prof_pool.back();
        g_prof_pool.pop_back();
    } else {
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    }
    r.flops = flops;
    r.bytes = bytes;
    r.kernel[0] = 0;
    strncpy(r.tag, tag ? tag : "", sizeof(r.tag) - 1);
    r.tag[sizeof(r.tag) - 1] = 0;
    (void)hipEventRecord(r.e0, s);
    g_prof_recs.push_back(r);
    slot = (int)g_prof_recs.size() - 1;
}

ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (slot < (int)g_prof_recs.size()) {
        (void)hipEventRecord(g_prof_recs[slot].e1, stream);
        strncpy(g_prof_recs[slot].kernel, g_prof_kernel, sizeof(g_prof_recs[slot].kernel) - 1);
        g_prof_recs[slot].kernel[sizeof(g_prof_recs[slot].kernel) - 1] = 0;
    }
}

}  // namespace bg

extern "C" {

int bg_abi_version(void) { return BG_ABI_VERSION; }
const char* bg_last_error(void) { return bg::g_err; }
const char* bg_target_arch(void) { return "gfx950"; }

// Host helper of the input pipeline (data.py): reverse the per-row PNG filters (ISO/IEC 15948 section 9) of an
// 8-bit image.  raw = h rows of (1 filter byte + stride data bytes); out = h * stride bytes.  Pure C on the CPU:
// the Sub / Average / Paeth filters are sequential along a row, which a Python loop decodes at ~20 images/s.
int bg_png_unfilter(const unsigned char* raw, int h, int stride, int bpp, unsigned char* out) {
    BG_REQUIRE(raw && out && h > 0 && stride > 0 && bpp > 0 && bpp <= 8, "bg_png_unfilter: bad argument");
    const unsigned char* prev = nullptr;
    for (int r = 0; r < h; ++r) {
        const unsigned char ft = raw[(size_t)r * (stride + 1)];
        const unsigned char* line = raw + (size_t)r * (stride + 1) + 1;
        unsigned char* cur = out + (size_t)r * stride;
        for (int i = 0; i < stride; ++i) {
            const int left = i >= bpp ? cur[i - bpp] : 0;
            const int up = prev ? prev[i] : 0;
            const int ul = (prev && i >= bpp) ? prev[i - bpp] : 0;
            int pred;
            switch (ft) {
                case 0: pred = 0; break;
                case 1: pred = left; break;
                case 2: pred = up; break;
                case 3: pred = (left + up) >> 1; break;
                case 4: {
                    const int p = left + up - ul;
                    const int pa = p > left ? p - left : left - p, pb = p > up ? p - up : up - p,
                              pc = p > ul ? p - ul : ul - p;
                    pred = (pa <= pb && pa <= pc) ? left : (pb <= pc ? up : ul);
                    break;
                }
                default: bg::set_error("bg_png_unfilter: bad filter type %d in row %d", (int)ft, r); return BG_ERR_ARG;
            }
            cur[i] = (unsigned char)((line[i] + pred) & 255);
        }
        prev = cur;
    }
    return BG_OK;
}

void bg_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(bg::g_prof_mu);
    bg::g_prof_on = on != 0;
}

void bg_prof_reset(void) {
    std::lock_guard<std::mutex> lk(bg::g_prof_mu);
    for (auto& r : bg::g_prof_recs) bg::g_prof_pool.push_back(r);
    bg::g_prof_recs.clear();
}

int bg_prof_dump(const char* path) {
    std::lock_guard<std::mutex> lk(bg::g_prof_mu);
    FILE* f = fopen(path, "w");
    if (!f) {
        bg::set_error("bg_prof_dump: cannot open %s", path);
        return BG_ERR_ARG;
    }
    fprintf(f, "tag\tkernel\tbytes\tflops\tms\n");     // tab-separated: kernel names carry commas
    for (auto& r : bg::g_prof_recs) {
        if (hipEventSynchronize(r.e1) != hipSuccess) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) continue;
        fprintf(f, "%s\t%s\t%.0f\t%.0f\t%.6f\n", r.tag, r.kernel, r.bytes, r.flops, (double)t);
    }
    fclose(f);
    return BG_OK;
}

int bg_prof_collect(double* total_ms, double* total_flops, int64_t* launches) {
    std::lock_guard<std::mutex> lk(bg::g_prof_mu);
    double ms = 0, fl = 0;
    int64_t n = 0;
    for (auto& r : bg::g_prof_recs) {
        if (hipEventSynchronize(r.e1) != hipSuccess) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) continue;
        ms += t;
        fl += r.flops;
        ++n;
    }
    for (auto& r : bg::g_prof_recs) bg::g_prof_pool.push_back(r);
    bg::g_prof_recs.clear();
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    if (launches) *launches = n;
    return BG_OK;
}

}  // extern "C"
