#!/bin/bash
# Diagnostic: price the parts of the decode kernels IN the 250-token loop by leaving them out (results wrong on purpose).
#   build (here, no GPU):   bash tests/diag/ar_ablate.sh build
#   run (GPU box):          bash tests/diag/ar_ablate.sh run > gpurun_out/ar_ablate.log
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
CS=$ROOT/tortoise_tts_amd/csrc
VARIANTS="1 2 3 4 8 16 32 64 128 256 320 60"
if [ "$1" = build ]; then
	make -C $CS -j8 > /dev/null
	for v in $VARIANTS; do
		mkdir -p $CS/build/abl$v
		for f in skinny attn; do
			/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DTTK_ABL=$v -c $CS/$f.hip -o $CS/build/abl$v/$f.o &
		done
	done
	wait
	for v in $VARIANTS; do
		objs=""
		for o in $CS/build/*.o; do
			b=$(basename $o .o)
			if [ $b = skinny ] || [ $b = attn ]; then objs="$objs $CS/build/abl$v/$b.o"; else objs="$objs $o"; fi
		done
		/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $ROOT/tortoise_tts_amd/libttk_abl$v.so
	done
	ls -la $ROOT/tortoise_tts_amd/libttk_abl*.so
else
	cd $ROOT
	python tests/diag/ar_ab.py 3 2>/dev/null
	for v in $VARIANTS; do
		TTK_LIB=$ROOT/tortoise_tts_amd/libttk_abl$v.so timeout -k 10 120 python tests/diag/ar_ab.py 3 2>/dev/null
	done
	python tests/diag/ar_ab.py 3 2>/dev/null
fi
