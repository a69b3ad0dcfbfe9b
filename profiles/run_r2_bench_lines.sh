R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final_r2
mkdir -p $O
cd $R
python bench.py > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --config c2 > $O/bench_c2.json 2> $O/bench_c2.err
python bench.py --config c5 > $O/bench_c5.json 2> $O/bench_c5.err
python bench.py --config c1 > $O/bench_c1.json 2> $O/bench_c1.err
for f in c3 c2 c5 c1; do tail -1 $O/bench_$f.json | cut -c1-260; done
