#!/bin/bash
# imagen-pytorch_amd/libimagen_hip_remat.so = the product sources with igemm.hip compiled -DIGEMM_EPI_REMAT.
exec bash "$(dirname "$0")/build_variant_lib.sh" remat -DIGEMM_EPI_REMAT
