# What-if timing of the backward's parts (variants built by tools/dbg/build_variant.sh; results of the non-base builds are WRONG on purpose)
cd $GRAFT_REPO_ROOT
./tools/ubench/systolic_parts > gpurun_out/systolic_parts.txt 2>&1
bash tools/dbg/ab_libs.sh > gpurun_out/whatif_bwd.txt 2>&1
