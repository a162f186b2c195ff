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

struct BlobTensor {
    const float* data;
    size_t numel;
    int ndim;
    uint32_t dims[8];
};

// TDXW container (targetdiarization_amd/weights.py:pack_blob):
//   "TDXW0001" | u32 n | n x { u16 name_len | name | u8 ndim | u32 dims[ndim] | u64 offset }
//   | zero pad to 64 | data section (f32 little-endian, each tensor 64-byte aligned)
struct Blob {
    std::map<std::string, BlobTensor> t;
    bool parse(const void* p, size_t bytes) {
        const uint8_t* b = (const uint8_t*)p;
        if (bytes < 12 || memcmp(b, "TDXW0001", 8) != 0) return false;
        size_t pos = 8;
        uint32_t n;
        memcpy(&n, b + pos, 4); pos += 4;
        struct Ent { std::string name; BlobTensor bt; uint64_t off; };
        std::vector<Ent> ents;
        ents.reserve(n);
        for (uint32_t i = 0; i < n; ++i) {
            if (pos + 2 > bytes) return false;
            uint16_t nl;
            memcpy(&nl, b + pos, 2); pos += 2;
            if (pos + nl + 1 > bytes) return false;
            Ent e;
            e.name.assign((const char*)b + pos, nl); pos += nl;
            e.bt.ndim = b[pos]; pos += 1;
            if (e.bt.ndim > 8 || pos + 4u * e.bt.ndim + 8 > bytes) return false;
            e.bt.numel = 1;
            for (int d = 0; d < e.bt.ndim; ++d) { memcpy(&e.bt.dims[d], b + pos, 4); pos += 4; e.bt.numel *= e.bt.dims[d]; }
            memcpy(&e.off, b + pos, 8); pos += 8;
            ents.push_back(e);
        }
        const size_t data0 = (pos + 63) / 64 * 64;
        for (auto& e : ents) {
            if (data0 + e.off + e.bt.numel * 4 > bytes) return false;
            e.bt.data = (const float*)(b + data0 + e.off);
            t[e.name] = e.bt;
        }
        return true;
    }
    const BlobTensor* find(const std::string& name) const {
        auto it = t.find(name);
        return it == t.end() ? nullptr : &it->second;
    }
};

}  // namespace tdx
