set -x
cd $GRAFT_REPO_ROOT
bash tools/profile_gpu.sh r04 > gpurun_out/prof_r04.log 2>&1
bash tools/profile_gpu.sh r04m3 --sampler-mode 3 > gpurun_out/prof_r04m3.log 2>&1
bash tools/profile_sq.sh r04sq > gpurun_out/prof_r04sq.log 2>&1
bash tools/profile_sq.sh r04sqm3 --sampler-mode 3 > gpurun_out/prof_r04sqm3.log 2>&1
bash tools/profile_train_traffic.sh r04train > gpurun_out/prof_r04train.log 2>&1
bash tools/profile_sq_train.sh r04sqtrain > gpurun_out/prof_r04sqtrain.log 2>&1
python tools/exp/eps_stress.py > gpurun_out/eps_stress_r04.txt 2>&1
python tools/bench_latency.py > gpurun_out/latency_r04.txt 2>&1
python bench.py > gpurun_out/bench_r04.json 2> gpurun_out/bench_r04.err
python bench.py --mode train > gpurun_out/bench_train_r04.json 2> gpurun_out/bench_train_r04.err
python bench.py --sampler-mode 3 --no-extras > gpurun_out/bench_r04_mode3.json 2> gpurun_out/bench_r04_mode3.err
tail -3 gpurun_out/latency_r04.txt
