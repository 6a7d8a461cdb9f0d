O=gpurun_out/final_r04; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3 > $O/pytest_gpu.txt
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -1 > $O/smoke.txt
timeout 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 200 python tools/ab_k3.py --sets corr,compact,minimal,dense_xyz,dense --steps 30 --rounds 3 > $O/ab_k3.txt 2>&1
timeout 200 python tools/ab_scannet.py --steps 30 > $O/ab_scannet.txt 2>&1
cat $O/pytest_gpu.txt $O/smoke.txt $O/ab_k3.txt $O/ab_scannet.txt; tail -c 400 $O/bench_n1.json
