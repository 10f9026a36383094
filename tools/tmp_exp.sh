python tools/splat_times.py 2>&1 | tail -2
timeout 300 python tools/quick_bench.py --pt --iters 3 2>&1 | tail -1
timeout 200 python tools/quick_bench.py --lvc --iters 2 --paths 1024 --vpl-paths 64 2>&1 | tail -1
rm -f build/kernels_splat.o; make -j8 all EXTRA_HIPFLAGS=-DEVPLP_NO_SPLAT_STATS >/dev/null 2>&1
echo "== no splat stats atomic"; python tools/splat_times.py 2>&1 | tail -2
