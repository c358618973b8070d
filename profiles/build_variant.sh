#!/bin/bash
# usage: bash profiles/build_variant.sh NAME "-DFLAG ..." [file.hip ...]
# Builds attention-gan_amd/csrc/variants/libagan_NAME.so: the shipped objects with the listed sources (default conv_p16.hip) recompiled
# under the extra flags -- diagnostic / ablation builds for `AGAN_LIB=.../libagan_NAME.so python profiles/conv_micro.py ...` (never shipped).
set -e
NAME=$1; FLAGS=$2; shift 2 || true
SRCS=${@:-conv_p16.hip}
cd "$(dirname "$0")/../attention-gan_amd/csrc"
make -j8 > /dev/null   # the shared objects must be current (headers change struct layouts)
mkdir -p variants
OBJS=""
for o in conv conv_patch conv_p16 conv_wgrows conv_wino conv_small bn_act attention damsm heads comm; do
  if echo " $SRCS " | grep -q " $o.hip "; then
    EXTRA=""; [ "$o" = conv_small ] && EXTRA="-fno-slp-vectorize"      # (the shipped per-file flags: csrc/Makefile)
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops $EXTRA $FLAGS -c $o.hip -o variants/${o}_$NAME.o \
        2> >(grep -v "is not a recognized feature for this target" >&2)
    OBJS="$OBJS variants/${o}_$NAME.o"
  else
    OBJS="$OBJS $o.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libagan_$NAME.so $OBJS -ldl
rm -f variants/*_$NAME.o
echo built variants/libagan_$NAME.so
