#!/bin/bash
# Build the library of a git revision (or of the working tree: "wt") into variants/<name>.so for same-box A/B runs
# (NEUMA_HIP_LIB=variants/<name>.so).   bash tools/build_variant.sh name [rev|wt] [extra make args]
name=$1; rev=${2:-wt}; shift; shift
R=$(cd $(dirname $0)/.. && pwd)
T=/tmp/variant_$name; rm -rf $T; mkdir -p $T/neuma_amd $R/variants
if [ "$rev" = "wt" ]; then cp -r $R/neuma_amd/csrc $T/neuma_amd/csrc; cp -r $R/include $T/include
else (cd $R && git archive $rev neuma_amd/csrc include) | tar -x -C $T; fi
rm -rf $T/neuma_amd/csrc/build
make -C $T/neuma_amd/csrc -j8 "$@" > $T/make.log 2>&1 || { tail -20 $T/make.log; exit 1; }
cp $T/neuma_amd/lib/libneuma_hip.so $R/variants/$name.so && echo "variants/$name.so"
