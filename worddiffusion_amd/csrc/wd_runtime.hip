// hipGraph capture helpers, per-kernel-class hipEvent timing, device info.
#include <string.h>

#include <vector>

#include "wd_common.h"

namespace {
struct ProfRec {
    int cls;
    double flops;
    hipEvent_t e0, e1;
};
int g_prof_on = 0;
std::vector<ProfRec> g_recs;
}  // namespace

extern "C" int wd_prof_is_on() { return g_prof_on; }

void wd_prof_begin(int cls, hipStream_t s, double flops) {
    ProfRec r;
    r.cls = cls;
    r.flops = flops;
    (void)hipEventCreate(&r.e0);
    (void)hipEventCreate(&r.e1);
    (void)hipEventRecord(r.e0, s);
    g_recs.push_back(r);
}
void wd_prof_end(hipStream_t s) {
    if (!g_recs.empty()) (void)hipEventRecord(g_recs.back().e1, s);
}

extern "C" int wd_prof_enable(int on) {
    g_prof_on = on ? 1 : 0;
    return WD_OK;
}

extern "C" int wd_prof_collect_flops(double* ms_per_class, int64_t* launches_per_class, double* flops_per_class) {
    if (!ms_per_class || !launches_per_class) return WD_EINVAL;
    if (hipDeviceSynchronize() != hipSuccess) return WD_ELAUNCH;
    for (int i = 0; i < WD_NCLASS; ++i) {
        ms_per_class[i] = 0.0;
        launches_per_class[i] = 0;
        if (flops_per_class) flops_per_class[i] = 0.0;
    }
    for (auto& r : g_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess && r.cls >= 0 && r.cls < WD_NCLASS) {
            ms_per_class[r.cls] += ms;
            launches_per_class[r.cls] += 1;
            if (flops_per_class) flops_per_class[r.cls] += r.flops;
        }
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    g_recs.clear();
    return WD_OK;
}

extern "C" int wd_prof_collect(double* ms_per_class, int64_t* launches_per_class, double* gemm_flops) {
    double fl[WD_NCLASS];
    const int rc = wd_prof_collect_flops(ms_per_class, launches_per_class, fl);
    if (rc == WD_OK && gemm_flops) *gemm_flops = fl[WD_CLS_GEMM];
    return rc;
}

extern "C" int wd_graph_begin(void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!st) return WD_EINVAL;  // the legacy default stream cannot be captured
    return hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) == hipSuccess ? WD_OK : WD_ELAUNCH;
}

extern "C" int wd_graph_end(void* stream, void** graph_exec_out) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!st || !graph_exec_out) return WD_EINVAL;
    hipGraph_t g = nullptr;
    if (hipStreamEndCapture(st, &g) != hipSuccess || !g) return WD_ESTATE;
    hipGraphExec_t ge = nullptr;
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return WD_ELAUNCH;
    *graph_exec_out = reinterpret_cast<void*>(ge);
    return WD_OK;
}

extern "C" int wd_graph_launch(void* graph_exec, void* stream) {
    if (!graph_exec) return WD_EINVAL;
    return hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(graph_exec), reinterpret_cast<hipStream_t>(stream)) ==
                   hipSuccess
               ? WD_OK
               : WD_ELAUNCH;
}

extern "C" int wd_graph_destroy(void* graph_exec) {
    if (!graph_exec) return WD_EINVAL;
    return hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(graph_exec)) == hipSuccess ? WD_OK : WD_ELAUNCH;
}

extern "C" const char* wd_version(void) { return "wdiff_hip 0.1 (gfx950)"; }

extern "C" int wd_device_info(int* cu_count, int* lds_per_block, char* name, int name_len) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return WD_ELAUNCH;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return WD_ELAUNCH;
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (lds_per_block) *lds_per_block = (int)p.sharedMemPerBlock;
    if (name && name_len > 0) {
        strncpy(name, p.gcnArchName, name_len - 1);
        name[name_len - 1] = 0;
    }
    return WD_OK;
}
