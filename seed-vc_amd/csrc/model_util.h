// Host-side helpers shared by the model packers: device allocations owned by a model, state-dict lookup,
// weight-norm folding and packing into the tap-GEMM weight layout ([Npad][Ktot], K-contiguous).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "../../include/seedvc_hip.h"
#include "common.h"
#include "kernels.h"

namespace svc {

struct Arena {
    std::vector<void*> ptrs;
    size_t total = 0;
    ~Arena() { release(); }
    void release() {
        for (void* p : ptrs) (void)hipFree(p);
        ptrs.clear();
        total = 0;
    }
    // zero-initialised device allocation; returns nullptr on failure (error already set)
    void* alloc(size_t bytes, hipStream_t st) {
        if (bytes == 0) bytes = 16;
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) {
            set_error("hipMalloc failed for " + std::to_string(bytes) + " bytes");
            return nullptr;
        }
        if (hipMemsetAsync(p, 0, bytes, st) != hipSuccess) {
            set_error("hipMemsetAsync failed");
            (void)hipFree(p);
            return nullptr;
        }
        ptrs.push_back(p);
        total += bytes;
        return p;
    }
    template <typename T>
    T* alloc_n(size_t n, hipStream_t st) { return reinterpret_cast<T*>(alloc(n * sizeof(T), st)); }
};

// Handle-owned pinned host staging for the small per-call host arrays (lengths, time grid).  A copy from pageable memory
// would have to be followed by a stream synchronisation before the caller's array may go away; copies from these slots
// are truly asynchronous.  A slot is reused only after the event recorded behind its last copies has completed
// (normally long ago: the wait is a no-op in steady state).
struct PinnedRing {
    static constexpr int NSLOT = 8;
    char* base = nullptr;
    size_t slot_bytes = 0;
    hipEvent_t ev[NSLOT] = {};
    bool pending[NSLOT] = {};
    int cur = 0;
    ~PinnedRing() {
        for (int i = 0; i < NSLOT; ++i) if (ev[i]) (void)hipEventDestroy(ev[i]);
        if (base) (void)hipHostFree(base);
    }
    // host pointer to a slot of at least `bytes`; nullptr on failure (error set)
    void* acquire(size_t bytes) {
        if (bytes > slot_bytes) {
            for (int i = 0; i < NSLOT; ++i) if (pending[i]) { (void)hipEventSynchronize(ev[i]); pending[i] = false; }
            if (base) (void)hipHostFree(base);
            slot_bytes = (bytes + 4095) & ~(size_t)4095;
            if (hipHostMalloc(reinterpret_cast<void**>(&base), slot_bytes * NSLOT, hipHostMallocDefault) != hipSuccess) {
                base = nullptr; slot_bytes = 0;
                set_error("hipHostMalloc failed for the staging ring");
                return nullptr;
            }
        }
        cur = (cur + 1) % NSLOT;
        if (!ev[cur] && hipEventCreateWithFlags(&ev[cur], hipEventDisableTiming) != hipSuccess) { set_error("hipEventCreate failed"); return nullptr; }
        if (pending[cur]) { (void)hipEventSynchronize(ev[cur]); pending[cur] = false; }
        return base + (size_t)cur * slot_bytes;
    }
    // call after the last hipMemcpyAsync that reads the current slot
    int commit(hipStream_t st) {
        if (hipEventRecord(ev[cur], st) != hipSuccess) { set_error("hipEventRecord failed"); return 1; }
        pending[cur] = true;
        return 0;
    }
};

struct StateDict {
    std::map<std::string, const svc_tensor_desc_t*> m;
    StateDict(const svc_tensor_desc_t* w, int n) {
        for (int i = 0; i < n; ++i) m[w[i].name] = &w[i];
    }
    const svc_tensor_desc_t* get(const std::string& k) const {
        auto it = m.find(k);
        return it == m.end() ? nullptr : it->second;
    }
    bool has(const std::string& k) const { return m.count(k) != 0; }
    static long numel(const svc_tensor_desc_t* d) {
        long n = 1;
        for (int i = 0; i < d->ndim; ++i) n *= d->shape[i];
        return n;
    }
};

// A packed tap-GEMM weight: [Npad][ldw] in fp16 or fp32, bias fp32 [N] (may be null)
struct PackedW {
    void* w = nullptr;
    long ldw = 0;
    int N = 0;
    int dtype = 0;   // 0 f16, 1 f32
    float* bias = nullptr;
};

// Resolves "<prefix>.weight" or "<prefix>.weight_g/_v" into (source pointer, per-dim0 scale or null).
struct WeightSrc {
    const float* v = nullptr;
    const float* scale = nullptr;   // device [dim0] = g / ||v||, null when not weight-normed
    const svc_tensor_desc_t* desc = nullptr;
};

inline int resolve_weight(const StateDict& sd, const std::string& prefix, Arena& ar, hipStream_t st, WeightSrc* out) {
    if (const auto* d = sd.get(prefix + ".weight")) {
        out->v = d->data;
        out->scale = nullptr;
        out->desc = d;
        return 0;
    }
    const auto* g = sd.get(prefix + ".weight_g");
    const auto* v = sd.get(prefix + ".weight_v");
    if (!g || !v) {
        set_error("state_dict is missing " + prefix + ".weight (or weight_g/weight_v)");
        return 1;
    }
    const int rows = (int)v->shape[0];
    float* sc = ar.alloc_n<float>(rows, st);
    if (!sc) return 1;
    if (wn_scale_launch(g->data, v->data, rows, StateDict::numel(v) / rows, sc, st)) return 1;
    out->v = v->data;
    out->scale = sc;
    out->desc = v;
    return 0;
}

inline int require_shape(const svc_tensor_desc_t* d, const std::string& name, std::initializer_list<long> shp) {
    if (!d) {
        set_error("state_dict is missing " + name);
        return 1;
    }
    bool ok = d->ndim == (int)shp.size();
    int i = 0;
    for (long s : shp) {
        if (ok && d->shape[i] != s) ok = false;
        ++i;
    }
    if (!ok) {
        std::string got;
        for (int j = 0; j < d->ndim; ++j) got += std::to_string(d->shape[j]) + (j + 1 < d->ndim ? "," : "");
        set_error("shape mismatch for " + name + ": got (" + got + ")");
        return 1;
    }
    return 0;
}

// element size helper
inline size_t esize(int dtype) { return dtype == 0 ? 2 : 4; }

// Copy a strided fp32 3-D block into a packed weight/bias buffer of the given dtype.
inline int pack_any(int dtype, const float* src, void* dst, long dst_off, int n0, int n1, int n2, long s0, long s1,
                    long s2, long d0, long d1, long d2, const float* scale, hipStream_t st) {
    if (dtype == 0)
        return pack_f16_launch(src, reinterpret_cast<half_t*>(dst) + dst_off, n0, n1, n2, s0, s1, s2, d0, d1, d2, scale, st);
    return pack_f32_launch(src, reinterpret_cast<float*>(dst) + dst_off, n0, n1, n2, s0, s1, s2, d0, d1, d2, scale, st);
}

// k-tile granularity (elements) of the tap-GEMM for a dtype
inline int ktile_elems(int dtype) { return dtype == 0 ? 64 : 32; }

}  // namespace svc
