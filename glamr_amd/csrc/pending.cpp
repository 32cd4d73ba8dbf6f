// Entry points declared in include/glamr_hip.h whose kernels are still being written: they fail loudly.
#include "common.hpp"
using namespace glamr;
#define PENDING(name) return fail(GLAMR_E_UNSUPPORTED, #name " is not implemented yet")
extern "C" int glamr_nets_create(glamr_nets**, const float*, const glamr_tensor_desc*, int, const float*, const glamr_tensor_desc*, int,
                                 const float*, const int32_t*) { PENDING(glamr_nets_create); }
extern "C" int glamr_nets_destroy(glamr_nets*) { return GLAMR_OK; }
extern "C" size_t glamr_nets_workspace_bytes(const glamr_nets*, int, int) { return 0; }
extern "C" int glamr_nets_infer(glamr_nets*, int, int, const int32_t*, const float*, const float*, const float*, int, const float*, float*,
                                float*, float*, float*, void*, void*) { PENDING(glamr_nets_infer); }
