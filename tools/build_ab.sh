#!/bin/bash
# Builds A/B variant libraries in the build container: bash tools/build_ab.sh "<tag>:<defines>" ...  -> tools/_ab_libs/libgpcc_<tag>.so
# (tag "default" = the product library, no defines).  Variants are built side by side; objects are cached per library.
cd "$(dirname "$0")/.."
mkdir -p tools/_ab_libs
for v in "$@"; do
  tag=${v%%:*}; defs=${v#*:}; [ "$defs" = "$v" ] && defs=""
  if [ "$tag" = default ]; then
    python3 -c "import sys; sys.path.insert(0,'.'); from gpcc_amd import build; print(build.build())" > /tmp/ab_$tag.log 2>&1 &
  else
    GPCC_BUILD_DEFINES="$defs" GPCC_HIP_LIB=$PWD/tools/_ab_libs/libgpcc_$tag.so python3 -c "import sys; sys.path.insert(0,'.'); from gpcc_amd import build; print(build.build())" > /tmp/ab_$tag.log 2>&1 &
  fi
done
wait
for v in "$@"; do tag=${v%%:*}; echo "== $tag"; tail -n 2 /tmp/ab_$tag.log; done
