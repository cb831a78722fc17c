#!/bin/bash
# build_variant.sh NAME [sed-script-file | -e 'sed expr' ...] : copy csrc to a scratch dir, apply the sed edits to ggs_render.hip
# (or the file named by VARIANT_FILE), build, and keep the library as csrc/variants/NAME.so (A/B timing via GGS_LIB_PATH,
# tools/dbg/job.sh ab).  What-if builds may compute WRONG results on purpose; they are never the product library.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; shift
SRC=$ROOT/gaussian-garments_amd/csrc
TMP=$(mktemp -d /tmp/ggsvar.XXXXXX)
mkdir -p $TMP/gaussian-garments_amd/csrc $TMP/include
cp $SRC/*.hip $SRC/*.h $SRC/Makefile $TMP/gaussian-garments_amd/csrc/
cp $ROOT/include/*.h $TMP/include/
F=${VARIANT_FILE:-ggs_render.hip}
# PATCHES="tools/dbg/variants/x.patch ...": unified diffs (paths relative to the repo root) applied to the scratch copy first --
# the round-4 reduction variants and what-if branches live there, outside the product translation unit
for pf in $PATCHES; do (cd $TMP && patch -p0 -s < $ROOT/$pf) || { echo "patch $pf failed"; exit 1; }; done
if [ $# -gt 0 ]; then sed -i "$@" $TMP/gaussian-garments_amd/csrc/$F; fi
make -C $TMP/gaussian-garments_amd/csrc -j8 RENDER_EXTRA="$RENDER_EXTRA" PERGAUSS_EXTRA="$PERGAUSS_EXTRA" ALL_EXTRA="$ALL_EXTRA" > $TMP/build.log 2>&1 || { tail -30 $TMP/build.log; exit 1; }
mkdir -p $SRC/variants
cp $TMP/gaussian-garments_amd/csrc/libggsplat.so $SRC/variants/$NAME.so
diff <(cat $SRC/$F) $TMP/gaussian-garments_amd/csrc/$F | head -40 || true
rm -rf $TMP
echo "built variants/$NAME.so"
