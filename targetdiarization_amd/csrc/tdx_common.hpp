// tdx_common.hpp — status/error plumbing and the TDXW weight-blob reader shared by the
// C-ABI translation units.  No exception crosses the C boundary: entry points return a
// TDX_E_* code and leave a message in a thread-local string (include/tdx.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace tdx {

inline std::string& last_error() {
    static thread_local std::string e;
    return e;
}
inline int fail(int code, const std::string& msg) {
    last_error() = msg;
    return code;
}
inline int fail_hip(hipError_t e, const char* file, int line) {
    last_error() = std::string("HIP error: ") + hipGetErrorString(e) + " at " + file + ":" + std::to_string(line);
    return 3;  // TDX_E_HIP
}

// Makes `device` current for the lifetime of the guard and restores the caller's device afterwards (the
// C-ABI never leaves the calling thread on another device; a forward on a handle whose device is not current
// launches on the handle's device — the stream the caller passes must belong to it).
struct DeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) err = hipSetDevice(device);
        else prev = -1;                 // nothing to restore
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

struct BlobTensor {
    const float* data;
    size_t numel;
    int ndim;
    uint32_t dims[8];
    mutable bool used = false;
};

// TDXW container (targetdiarization_amd/weights.py:pack_blob):
//   "TDXW0001" | u32 n | n x { u16 name_len | name | u8 ndim | u32 dims[ndim] | u64 offset }
//   | zero pad to 64 | data section (f32 little-endian, each tensor 64-byte aligned)
// The header is untrusted input: every count, product and offset is checked without wrap-around.
struct Blob {
    std::map<std::string, BlobTensor> t;
    bool parse(const void* p, size_t bytes) {
        const uint8_t* b = (const uint8_t*)p;
        if (!b || bytes < 12 || memcmp(b, "TDXW0001", 8) != 0) return false;
        size_t pos = 8;
        uint32_t n;
        memcpy(&n, b + pos, 4); pos += 4;
        if ((size_t)n > (bytes - pos) / 11) return false;            // an entry is at least 2 + 0 + 1 + 0 + 8 bytes
        struct Ent { std::string name; BlobTensor bt; uint64_t off; };
        std::vector<Ent> ents;
        ents.reserve(n);
        for (uint32_t i = 0; i < n; ++i) {
            if (bytes - pos < 2) return false;
            uint16_t nl;
            memcpy(&nl, b + pos, 2); pos += 2;
            if (bytes - pos < (size_t)nl + 1) return false;
            Ent e;
            e.name.assign((const char*)b + pos, nl); pos += nl;
            e.bt.ndim = b[pos]; pos += 1;
            if (e.bt.ndim > 8 || bytes - pos < 4u * (size_t)e.bt.ndim + 8) return false;
            uint64_t numel = 1;
            for (int d = 0; d < e.bt.ndim; ++d) {
                memcpy(&e.bt.dims[d], b + pos, 4); pos += 4;
                if (e.bt.dims[d] != 0 && numel > (uint64_t)bytes / e.bt.dims[d]) return false;   // numel*4 could not fit anyway
                numel *= e.bt.dims[d];
            }
            e.bt.numel = (size_t)numel;
            memcpy(&e.off, b + pos, 8); pos += 8;
            ents.push_back(e);
        }
        if (bytes - pos < (64 - pos % 64) % 64) return false;
        const size_t data0 = (pos + 63) / 64 * 64;
        const size_t room = bytes - data0;
        for (auto& e : ents) {
            if (e.off % 4 != 0 || e.off > room) return false;
            if (e.bt.numel > (room - (size_t)e.off) / 4) return false;
            e.bt.data = (const float*)(b + data0 + e.off);
            if (t.count(e.name)) return false;                       // duplicate name
            t[e.name] = e.bt;
        }
        return true;
    }
    const BlobTensor* find(const std::string& name) const {
        auto it = t.find(name);
        if (it == t.end()) return nullptr;
        it->second.used = true;
        return &it->second;
    }
    // first tensor no find() asked for ("" if none): strict loaders reject unexpected keys like
    // load_state_dict(strict=True) does (base_model.py:63)
    std::string first_unused() const {
        for (const auto& kv : t) if (!kv.second.used) return kv.first;
        return std::string();
    }
};

}  // namespace tdx
