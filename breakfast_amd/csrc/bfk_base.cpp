// bfk_base.cpp — the HIP-free base of the C-ABI (libbfk_front.so, which libbfk.so links): error state, bfk_free,
// and the PRELOAD thread.  A CLI run spends ~0.18 s loading the HIP runtime, creating the device context and loading
// the code object before its first kernel; none of it depends on the input.  bfk_preload_start() does all of that on
// a native thread (dlopen of libbfk.so + bfk_warmup) while the caller reads and tokenises the input with the
// text stages of this library; bfk_table_cluster_write() joins the thread and hands the table's CSR to
// bfk_cluster_csr.  Without a preload the same calls simply happen in line.
#include "../../include/bfk.h"

#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

static thread_local std::string g_err;

int bfk_fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

extern "C" int bfk_abi_version(void) { return BFK_ABI_VERSION; }
extern "C" const char *bfk_last_error(void) { return g_err.c_str(); }
extern "C" void bfk_free(void *p) { free(p); }

namespace {
std::mutex g_mu;
// The thread is never left joinable at static destruction (std::terminate): the holder's destructor joins it, and every
// caller that started it joins earlier — bfk_preload_wait on the clustering path, bfk_preload_join from the Python shell's
// atexit hook on every other exit (max-dist 0, declined inputs, exceptions), i.e. before the HIP runtime's own exit
// handlers run.
struct ThreadHolder {
    std::thread th;
    ~ThreadHolder() {
        if (th.joinable()) th.join();
    }
};
ThreadHolder g_thread;
bool g_started = false, g_joined = false;
void *g_handle = nullptr;
std::string g_path, g_load_err;
int g_warm_rc = 0;
std::string g_warm_err;

typedef int (*warmup_fn)(int, int64_t, int64_t);
typedef int (*cluster_fn)(const int32_t *, const int32_t *, int64_t, int32_t, int32_t, int32_t *, bfk_stats *);
typedef const char *(*err_fn)(void);

void preload_body(int device, int64_t rows_hint, int64_t nnz_hint) {
    g_handle = dlopen(g_path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!g_handle) {
        const char *e = dlerror();
        g_load_err = e ? e : "dlopen failed";
        return;
    }
    if (warmup_fn w = (warmup_fn)dlsym(g_handle, "bfk_warmup")) {
        g_warm_rc = w(device, rows_hint, nnz_hint);
        if (g_warm_rc)
            if (err_fn le = (err_fn)dlsym(g_handle, "bfk_last_error")) g_warm_err = le();  // (this thread's message)
    }
}
}  // namespace

extern "C" int bfk_preload_start(const char *libbfk_path, int device, int64_t rows_hint, int64_t nnz_hint) {
    if (!libbfk_path) return bfk_fail(BFK_EARG, "bfk_preload_start: null path");
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_started) return BFK_OK;
    g_path = libbfk_path;
    g_started = true;
    g_thread.th = std::thread(preload_body, device, rows_hint, nnz_hint);
    return BFK_OK;
}

// wait for the preload thread whatever it achieved (no error is reported): for exit paths that never cluster
extern "C" void bfk_preload_join(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_started && !g_joined) {
        if (g_thread.th.joinable()) g_thread.th.join();
        g_joined = true;
    }
}

extern "C" int bfk_preload_wait(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_started) return bfk_fail(BFK_ESTATE, "bfk_preload_wait: bfk_preload_start was not called");
    if (!g_joined) {
        if (g_thread.th.joinable()) g_thread.th.join();
        g_joined = true;
    }
    if (!g_handle) return bfk_fail(BFK_ENODEV, "cannot load " + g_path + ": " + g_load_err);
    if (g_warm_rc) return bfk_fail(g_warm_rc, g_warm_err);
    return BFK_OK;
}

// libbfk.so's bfk_cluster_csr through the preloaded handle (used by bfk_table_cluster_write in bfk_frontend.cpp)
int bfk_front_cluster(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int32_t max_dist, int32_t n_gpus,
                      int32_t *labels_out) {
    if (int rc = bfk_preload_wait()) return rc;
    cluster_fn f = (cluster_fn)dlsym(g_handle, "bfk_cluster_csr");
    if (!f) return bfk_fail(BFK_ENODEV, g_path + " does not export bfk_cluster_csr");
    return f(indptr, indices, n_rows, max_dist, n_gpus < 1 ? 1 : n_gpus, labels_out, nullptr);  // (errors: same thread, same bfk_last_error)
}

// libbfk.so's bfk_table_cluster_write_device (filter + collapse + CSR + clustering on the device, writer) through the preloaded handle
typedef int (*pipeline_fn)(bfk_table *, const char *, int64_t, const bfk_filter_opts *, int32_t, int32_t, const char *, bfk_prep_info *, int64_t *);
extern "C" int bfk_table_pipeline_device(bfk_table *t, const char *sep2, int64_t sep2_len, const bfk_filter_opts *opts, int32_t max_dist,
                                         int32_t min_cluster_size, const char *path, bfk_prep_info *info_out, int64_t *n_clusters_out) {
    const bool timing = getenv("BFK_FRONT_TIMING") && atoi(getenv("BFK_FRONT_TIMING")) != 0;
    const auto t0 = std::chrono::steady_clock::now();
    if (int rc = bfk_preload_wait()) return rc;
    if (timing)
        fprintf(stderr, "[bfk_dev] waited for the preload thread %8.2f ms\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    pipeline_fn f = (pipeline_fn)dlsym(g_handle, "bfk_table_cluster_write_device");
    if (!f) return bfk_fail(BFK_ENODEV, g_path + " does not export bfk_table_cluster_write_device");
    return f(t, sep2, sep2_len, opts, max_dist, min_cluster_size, path, info_out, n_clusters_out);  // (errors: same thread, same bfk_last_error)
}

// ... as a side-car cache run: an exact input cache checked, every edge recorded and a side-car written (bfk_table_cluster_write_device_cache)
typedef int (*pipeline_cache_fn)(bfk_table *, const char *, int64_t, const bfk_filter_opts *, int32_t, int32_t, const char *, const char *, const char *,
                                 bfk_prep_info *, int64_t *);
extern "C" int bfk_table_pipeline_device_cache(bfk_table *t, const char *sep2, int64_t sep2_len, const bfk_filter_opts *opts, int32_t max_dist,
                                               int32_t min_cluster_size, const char *path, const char *in_cache, const char *cache_path,
                                               bfk_prep_info *info_out, int64_t *n_clusters_out) {
    if (int rc = bfk_preload_wait()) return rc;
    pipeline_cache_fn f = (pipeline_cache_fn)dlsym(g_handle, "bfk_table_cluster_write_device_cache");
    if (!f) return bfk_fail(BFK_ENODEV, g_path + " does not export bfk_table_cluster_write_device_cache");
    return f(t, sep2, sep2_len, opts, max_dist, min_cluster_size, path, in_cache, cache_path, info_out, n_clusters_out);
}

// ... with the clustering on n_gpus devices where that pays (bfk_table_cluster_write_device_gpus)
typedef int (*pipeline_gpus_fn)(bfk_table *, const char *, int64_t, const bfk_filter_opts *, int32_t, int32_t, int32_t, const char *, bfk_prep_info *, int64_t *);
extern "C" int bfk_table_pipeline_device_gpus(bfk_table *t, const char *sep2, int64_t sep2_len, const bfk_filter_opts *opts, int32_t max_dist,
                                              int32_t min_cluster_size, int32_t n_gpus, const char *path, bfk_prep_info *info_out,
                                              int64_t *n_clusters_out) {
    if (int rc = bfk_preload_wait()) return rc;
    pipeline_gpus_fn f = (pipeline_gpus_fn)dlsym(g_handle, "bfk_table_cluster_write_device_gpus");
    if (!f) return bfk_fail(BFK_ENODEV, g_path + " does not export bfk_table_cluster_write_device_gpus");
    return f(t, sep2, sep2_len, opts, max_dist, min_cluster_size, n_gpus, path, info_out, n_clusters_out);
}
