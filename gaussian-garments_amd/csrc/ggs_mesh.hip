// ggs_mesh.hip -- placeholder until the fused mesh-binding kernels land (next commit).
#include "ggs_kernels.h"
extern "C" {
int ggs_mesh_bind_forward(int, int, const float*, const int64_t*, const int64_t*, const float*, const float*,
                          const float*, const float*, float*, float*, float*, void*) { return GGS_ERR_ARG; }
int ggs_mesh_bind_backward(int, int, const float*, const int64_t*, const int64_t*, const float*, const float*,
                           const float*, const float*, const float*, const float*, const float*, float*, float*,
                           float*, float*, void*) { return GGS_ERR_ARG; }
}
