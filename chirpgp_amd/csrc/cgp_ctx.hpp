// cgp_ctx.hpp -- the opaque context of include/chirpgp_hip.h, shared by the translation units that implement the C-ABI.
#pragma once
#include <string>

struct cgp_ctx {
    int device;
    int num_cus;
    std::string err;
};

namespace cgp {
inline int fail(cgp_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}
}  // namespace cgp
