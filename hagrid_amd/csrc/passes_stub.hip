// temporary: construction passes not built yet
#include "ctx.h"
using namespace hagrid_impl;
extern "C" int hagrid_build_grid(hagrid_ctx* ctx, const void*, int, hagrid_grid*, float, float) { HG_FAIL(ctx, HAGRID_EINVAL, "not implemented"); }
extern "C" int hagrid_merge_grid(hagrid_ctx* ctx, hagrid_grid*, float) { HG_FAIL(ctx, HAGRID_EINVAL, "not implemented"); }
extern "C" int hagrid_flatten_grid(hagrid_ctx* ctx, hagrid_grid*) { HG_FAIL(ctx, HAGRID_EINVAL, "not implemented"); }
extern "C" int hagrid_expand_grid(hagrid_ctx* ctx, hagrid_grid*, const void*, int) { HG_FAIL(ctx, HAGRID_EINVAL, "not implemented"); }
extern "C" int hagrid_compress_grid(hagrid_ctx* ctx, hagrid_grid*) { HG_FAIL(ctx, HAGRID_EINVAL, "not implemented"); }
