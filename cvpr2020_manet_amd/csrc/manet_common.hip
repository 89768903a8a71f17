// version / error plumbing of libmanet_hip.so
#include "manet_common.h"

static thread_local char g_manet_err[512] = "";

int manet_set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_manet_err, sizeof(g_manet_err), fmt, ap);
    va_end(ap);
    return code;
}

// ---- opt-in profiling: HIP event pairs around the dominant kernel (channel 0), the local-window stage (channel 1),
// the per-frame prepare kernel (channel 2) and the exact re-rank of the bf16-filter mode (channel 3) ------------
#include <mutex>
#include <stdlib.h>
#include <vector>
namespace {
constexpr int PROF_CH = 4;
struct ProfState {
    std::mutex mu;
    bool enabled = false;
    size_t used[PROF_CH] = {0, 0, 0, 0};  // events used per channel (2 per bracket)
    std::vector<hipEvent_t> ev[PROF_CH];
} g_prof;
}  // namespace

void manet_profile_record(hipStream_t st, bool start, int channel)
{
    if (!g_prof.enabled) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (!g_prof.enabled || channel < 0 || channel >= PROF_CH) return;
    size_t &used = g_prof.used[channel];
    std::vector<hipEvent_t> &ev = g_prof.ev[channel];
    if (start && used + 2 > ev.size()) return;  // pool exhausted: stop recording
    if (!start && (used & 1) == 0) return;       // no matching start
    (void)hipEventRecord(ev[used++], st);
}

static int g_tune[MANET_TUNE_COUNT] = {0};
static bool g_tune_set[MANET_TUNE_COUNT] = {false};
// experiments without touching a caller's code: MANET_TUNING=1 MANET_TUNE_INIT="11=128,12=17" sets knobs at the first read
static std::once_flag g_tune_env_once;
static void tune_env_init()
{
    const char *opt = getenv("MANET_TUNING"), *init = getenv("MANET_TUNE_INIT");
    if (!opt || opt[0] != '1' || !init) return;
    for (const char *p = init; *p;) {
        char *end = nullptr;
        const long key = strtol(p, &end, 10);
        if (end == p || *end != '=') break;
        p = end + 1;
        const long val = strtol(p, &end, 10);
        if (end == p) break;
        if (key >= 0 && key < MANET_TUNE_COUNT) {
            g_tune[key] = (int)val;
            g_tune_set[key] = true;
        }
        p = (*end == ',' || *end == ';') ? end + 1 : end;
        if (*end != ',' && *end != ';') break;
    }
}
int manet_tune_get(int key, int dflt)
{
    std::call_once(g_tune_env_once, tune_env_init);
    return (key >= 0 && key < MANET_TUNE_COUNT && g_tune_set[key]) ? g_tune[key] : dflt;
}

extern "C" {

int manet_tune_set(int key, int value)
{
    // experiments only: the data path reads these knobs, so the "no state between calls" promise of manet_hip.h holds
    // only while nobody sets them -- refuse unless the process opted in
    const char *opt = getenv("MANET_TUNING");
    if (!opt || opt[0] != '1')
        return manet_set_error(MANET_E_INVALID, "manet_tune_set is for experiments: set MANET_TUNING=1 in the environment");
    if (key < 0 || key >= MANET_TUNE_COUNT) return manet_set_error(MANET_E_INVALID, "tune key %d", key);
#ifndef MANET_ABLATION
    // the default build carries only the kernels the data path can reach; the variant / ablation instantiations these keys
    // select are compiled in by `make EXTRA=-DMANET_ABLATION`
    if (key == MANET_TUNE_BF16_VARIANT || key == MANET_TUNE_ABLATION || key == MANET_TUNE_F32_UNPIPED || key == MANET_TUNE_FRAME_XC)
        return manet_set_error(MANET_E_INVALID, "tune key %d selects kernel variants this library was built without: rebuild "
                                                "with `make -C cvpr2020_manet_amd/csrc clean all EXTRA=-DMANET_ABLATION`", key);
#endif
    g_tune[key] = value;
    g_tune_set[key] = value != INT32_MIN;  // INT32_MIN: back to "not set" (the shipped default, e.g. the automatic block map)
    return MANET_OK;
}

int manet_profile_begin(int max_launches)
{
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (max_launches <= 0) return manet_set_error(MANET_E_INVALID, "max_launches=%d", max_launches);
    for (int c = 0; c < PROF_CH; ++c) {
        for (hipEvent_t e : g_prof.ev[c]) (void)hipEventDestroy(e);
        g_prof.ev[c].assign((size_t)max_launches * 2, nullptr);
        for (auto &e : g_prof.ev[c])
            if (hipEventCreate(&e) != hipSuccess) return manet_set_error(MANET_E_NODEVICE, "hipEventCreate failed");
        g_prof.used[c] = 0;
    }
    g_prof.enabled = true;
    return MANET_OK;
}

static int profile_collect(int c, float *ms_out, int capacity, int *n_launches)
{
    int n = (int)(g_prof.used[c] / 2);
    int wrote = 0;
    for (int i = 0; i < n && i < capacity; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(g_prof.ev[c][2 * i + 1]) != hipSuccess ||
            hipEventElapsedTime(&ms, g_prof.ev[c][2 * i], g_prof.ev[c][2 * i + 1]) != hipSuccess)
            return manet_set_error(MANET_E_LAUNCH, "event timing failed");
        if (ms_out) ms_out[i] = ms;
        ++wrote;
    }
    if (n_launches) *n_launches = wrote;
    return MANET_OK;
}

int manet_profile_read(int channel, float *ms_out, int capacity, int *n_launches)
{
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (channel < 0 || channel >= PROF_CH) return manet_set_error(MANET_E_INVALID, "profile channel %d", channel);
    return profile_collect(channel, ms_out, capacity, n_launches);
}

int manet_profile_end2(float *ms_out, int capacity, int *n_launches, float *local_ms_out, int local_capacity,
                       int *n_local)
{
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.enabled = false;
    int rc = profile_collect(0, ms_out, capacity, n_launches);
    if (!rc) rc = profile_collect(1, local_ms_out, local_capacity, n_local);
    for (int c = 0; c < PROF_CH; ++c) {
        for (hipEvent_t e : g_prof.ev[c]) (void)hipEventDestroy(e);
        g_prof.ev[c].clear();
        g_prof.used[c] = 0;
    }
    return rc;
}

int manet_profile_end(float *ms_out, int capacity, int *n_launches)
{
    return manet_profile_end2(ms_out, capacity, n_launches, nullptr, 0, nullptr);
}

const char *manet_version(void) { return "manet_hip 0.1 (gfx950)"; }
const char *manet_last_error_string(void) { return g_manet_err; }
}
