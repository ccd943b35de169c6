#!/bin/bash
# round 3, job i: two gated microbenchmarks -- the Shoup (precomputed-quotient) twiddle product against the Montgomery asm block,
# and the int8-MFMA route for the constant half of the Montgomery reduction
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R/tools/microbench
timeout 300 ./shoup > $O/r03i_shoup_product.txt 2>&1
timeout 300 ./mfma_mp_bound > $O/r03i_mfma_mp_bound.txt 2>&1
cat $O/r03i_shoup_product.txt $O/r03i_mfma_mp_bound.txt
