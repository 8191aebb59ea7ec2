// roctx ranges behind the C ABI ("Tracing / profiling": the reference has none).  The marker library is resolved
// at run time and only when SDNET_ROCTX=1, so an untraced process pays one predictable branch per range and the library loads on hosts
// without a profiler.  `rocprofv3 --marker-trace --kernel-trace -- python3 bench.py` then shows render / forward stages / loss /
// backward stages / every gradient bucket / Adam as nested ranges on the launching thread.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include "../../include/sdnet_hip.h"

namespace sd {
void set_error(const char* fmt, ...);
typedef int (*fn_push)(const char*);
typedef int (*fn_pop)(void);
typedef void (*fn_mark)(const char*);
struct Roctx {
    fn_push push = nullptr;
    fn_pop pop = nullptr;
    fn_mark mark = nullptr;
    const char* soname = nullptr;
};
static Roctx g_roctx;
static std::once_flag g_roctx_once;

static const Roctx* roctx() {
    std::call_once(g_roctx_once, [] {
        const char* on = getenv("SDNET_ROCTX");
        if (!on || !*on || !strcmp(on, "0")) return;
        // rocprofv3 intercepts the rocprofiler-sdk marker library; libroctx64 is the roctracer-era one (rocprof v1 / v2)
        static const char* names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so",
                                      "/opt/rocm/lib/librocprofiler-sdk-roctx.so.1", "/opt/rocm/lib/libroctx64.so.4"};
        for (const char* n : names) {
            void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            Roctx r;
            r.push = (fn_push)dlsym(h, "roctxRangePushA");
            r.pop = (fn_pop)dlsym(h, "roctxRangePop");
            r.mark = (fn_mark)dlsym(h, "roctxMarkA");
            r.soname = n;
            if (r.push && r.pop) { g_roctx = r; return; }
        }
    });
    return g_roctx.push ? &g_roctx : nullptr;
}
}  // namespace sd

extern "C" {
int sd_range_enabled(void) { return sd::roctx() ? 1 : 0; }

const char* sd_range_library(void) {
    const sd::Roctx* r = sd::roctx();
    return r ? r->soname : "";
}

int sd_range_push(const char* name) {
    const sd::Roctx* r = sd::roctx();
    if (!r) return 0;
    if (!name) { sd::set_error("sd_range_push: null name"); return SD_ERR_INVALID; }
    r->push(name);
    return 0;
}

int sd_range_pop(void) {
    const sd::Roctx* r = sd::roctx();
    if (r) r->pop();
    return 0;
}

int sd_range_mark(const char* name) {
    const sd::Roctx* r = sd::roctx();
    if (r && r->mark && name) r->mark(name);
    return 0;
}
}
