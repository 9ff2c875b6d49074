"""`simple_knn._C.distCUDA2` (scene/gaussian_model.py:20,163-166 of the reference): mean squared distance of
every point to its 3 nearest neighbours, used once at scene creation to size the first Gaussians.  The
reference installs it from an absent third-party submodule (.gitmodules:1-3); here it is the HIP operator
`gsr_knn_mean_dist2` of libgsraster.so (include/gsraster.h): exact 3-NN, Morton-ordered boxes, gfx950 only --
there is no CPU path."""
from diff_gaussian_rasterization import knn_mean_dist2


def distCUDA2(points):
    return knn_mean_dist2(points)
