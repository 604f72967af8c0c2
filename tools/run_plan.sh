for t in 1 4 8 16 32; do
 echo "plan_threads $t"; TSGPU_HOST_TIMING=1 python bench.py --workload keyword --no-extras --no-cpu-baseline --steps 3 --warmup 1 --opt plan_threads=$t 2>&1 | grep -E 'plan:|kw batch 10000' | tail -2
done
