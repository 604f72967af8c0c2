#!/bin/bash
# The profiling recipe behind profiles/rNN/ (run on the GPU box through gpurun):  tools/gpu_profile.sh r02 [tag]
#   * rocprofv3 --kernel-trace --stats of the keyword and the vector bench legs       -> rocprof_{keyword,vector}_<tag>_stats.txt
#   * PMC passes, each in its OWN run (no trace domains next to --pmc): FETCH_SIZE; SQ instruction mix; SQ wait / LDS conflicts
#     -> pmc_{kw,vec}_fetch.txt, pmc_kw_sq1.txt, pmc_kw_sq2.txt (per-dispatch averages by tools/pmc_summary.py)
#   * the GPU test tier, the smoke test and the full bench line                        -> pytest_gpu_<tag>.txt, smoke_<tag>.txt, bench_all_<tag>.json
# bench.py reads pmc_kw_fetch.txt / pmc_kw_sq1.txt / pmc_vec_fetch.txt of the round for roofline.traffic / issue_util.
set -u
R=${1:-r06}; TAG=${2:-final}
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out/prof_$TAG; P=$ROOT/gpurun_out/profiles_$R
mkdir -p $O $P
cd /tmp
KW="python $ROOT/bench.py --workload keyword --no-cpu-baseline --no-extras --steps 3 --warmup 1"
VEC="python $ROOT/bench.py --workload vector --no-cpu-baseline --no-extras --steps 3 --warmup 1"
KWG="python $ROOT/bench.py --workload kwgeneral --no-cpu-baseline --steps 3 --warmup 1"     # two query_by fields + 10 candidate combinations per query (general kernels)
if [ "${SKIP_PROF:-0}" != "1" ]; then          # SKIP_PROF=1: only the GPU tier, the smoke test and the bench line (the committed rocprof / PMC summaries stay)
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_kw -- $KW > $O/trace_kw.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_vec -- $VEC > $O/trace_vec.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_kwg -- $KWG > $O/trace_kwg.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_kw_fetch -- $KW > $O/pmc_kw_fetch.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_vec_fetch -- $VEC > $O/pmc_vec_fetch.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d $O/pmc_kw_sq1 -- $KW > $O/pmc_kw_sq1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU -d $O/pmc_kw_sq2 -- $KW > $O/pmc_kw_sq2.log 2>&1
# round 6: the general-kernel leg's counters, too (two query_by fields: kw_find_mf2_kernel + kw_score_kernel<.., MF>; group_by: gb_*; facets: facet_*)
timeout 900 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_kwg_fetch -- $KWG > $O/pmc_kwg_fetch.log 2>&1
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d $O/pmc_kwg_sq1 -- $KWG > $O/pmc_kwg_sq1.log 2>&1
timeout 900 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU -d $O/pmc_kwg_sq2 -- $KWG > $O/pmc_kwg_sq2.log 2>&1
cd $ROOT
python profiles/summarize_rocprof.py $O/trace_kw > $P/rocprof_keyword_${TAG}_stats.txt 2>&1
python profiles/summarize_rocprof.py $O/trace_vec > $P/rocprof_vector_${TAG}_stats.txt 2>&1
python profiles/summarize_rocprof.py $O/trace_kwg > $P/rocprof_kwgeneral_${TAG}_stats.txt 2>&1
python profiles/summarize_rocprof.py $O/trace_kwg gb_ > $P/rocprof_groupby_${TAG}_stats.txt 2>&1          # the group_by leg's kernels (kw_groupby.hip.h)
python tools/pmc_summary.py $O/pmc_kw_fetch "kw_" > $P/pmc_kw_fetch.txt 2>&1
python tools/pmc_summary.py $O/pmc_vec_fetch "vec_" > $P/pmc_vec_fetch.txt 2>&1
python tools/pmc_summary.py $O/pmc_kw_sq1 "kw_" > $P/pmc_kw_sq1.txt 2>&1
python tools/pmc_summary.py $O/pmc_kw_sq2 "kw_" > $P/pmc_kw_sq2.txt 2>&1
for c in fetch sq1 sq2; do python tools/pmc_summary.py $O/pmc_kwg_$c "kw_find_mf2|kw_score_kernel|kw_candidates|gb_|facet_" > $P/pmc_kwg_$c.txt 2>&1; done
python profiles/summarize_rocprof.py $O/trace_kwg facet_ > $P/rocprof_facets_${TAG}_stats.txt 2>&1
# which kernel sources these counters belong to (bench.py compares: roofline.issue_util.pmc_of_these_sources)
python - > $P/pmc_meta.json <<PY
import json, sys, hashlib
sys.path.insert(0, "$ROOT")
import bench
print(json.dumps({"kernel_src_sha16": bench.kernel_src_sha16(), "libtsgpu_sha16": hashlib.sha256(open("$ROOT/typesense_amd/libtsgpu.so", "rb").read()).hexdigest()[:16], "tag": "$TAG"}))
PY
fi
cd $ROOT
# the bench reads the round's PMC summaries from profiles/<round>/ : put them there for THIS run, too
mkdir -p profiles/$R && { ls $P/pmc_*.txt > /dev/null 2>&1 && cp $P/pmc_*.txt $P/pmc_meta.json profiles/$R/; }
timeout 1500 python -m pytest tests -m gpu -x -q > $P/pytest_gpu_$TAG.txt 2>&1; tail -2 $P/pytest_gpu_$TAG.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $P/smoke_$TAG.txt 2>&1; tail -1 $P/smoke_$TAG.txt
# (round 5: stdout = the compact line the driver parses; the full record = --detail-out; stderr carries a copy of it)
( time timeout 1200 python bench.py --steps 20 --warmup 5 --detail-out $P/bench_all_${TAG}_detail.json ) > $P/bench_all_$TAG.json 2> $O/bench_all.err; grep -v BENCH_DETAIL $O/bench_all.err | tail -4; wc -c $P/bench_all_$TAG.json $P/bench_all_${TAG}_detail.json
# round 6: the HNSW leg at BASELINE config 3's size (10M x 768; the graph built on the device by tsgpu_vec_hnsw_build) + the kernel trace of a 2M-row bulk build
if [ "${SKIP_HNSW10M:-0}" != "1" ]; then
( time timeout 1500 python bench.py --workload hnsw --no-cpu-baseline --hnsw-rows 10000000 ) > $P/bench_hnsw_10m.json 2> $O/bench_hnsw_10m.err; tail -3 $O/bench_hnsw_10m.err
( cd /tmp; HNSW_SKIP_INSERTED=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_hnsw -- python $ROOT/tools/experiments/exp_r06_hnsw_bulk.py 2000000 > $O/trace_hnsw.log 2>&1 )
python profiles/summarize_rocprof.py $O/trace_hnsw vec_hnsw > $P/rocprof_hnsw_build_${TAG}_stats.txt 2>&1
fi
rocm-smi --showmeminfo vram 2>/dev/null | head -5 > $P/hw_$TAG.txt; rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 >> $P/hw_$TAG.txt
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete; rm -rf $O/trace_* $O/pmc_*
du -sh $ROOT/gpurun_out/profiles_$R
