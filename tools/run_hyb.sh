for t in 16 32 64 128; do
 echo "fuse_threads $t"; TSGPU_HOST_TIMING=1 python bench.py --workload hybrid --no-extras --no-cpu-baseline --steps 4 --warmup 2 --opt fuse_threads=$t 2>&1 | grep -E 'hybrid batch' | tail -2
done
