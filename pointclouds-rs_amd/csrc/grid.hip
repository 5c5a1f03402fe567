// placeholder until the grid kernels land
#include "knn_common.h"
int pcrs_grid_build(pcrs_index *ix, uint32_t) { ix->algo = PCRS_ALGO_BRUTE; return PCRS_OK; }
int pcrs_launch_knn_grid(pcrs_index *ix, const KnnQuery &, const KnnOut &, uint32_t) {
    return pcrs_set_err(ix->ctx, PCRS_ERR_INVALID_ARG, "grid search not built");
}
