# Runs on the GPU box: every profile summary of round 5 (copied from gpurun_out/ into profiles/ afterwards)
set -x
cd $GRAFT_REPO_ROOT
bash tools/profile_gpu.sh r05m3 --sampler-mode 3 > gpurun_out/prof_r05m3.log 2>&1
bash tools/profile_gpu.sh r05m4 --sampler-mode 4 > gpurun_out/prof_r05m4.log 2>&1
bash tools/profile_sq.sh r05sqm3 --sampler-mode 3 > gpurun_out/prof_r05sqm3.log 2>&1
bash tools/profile_sq.sh r05sqm4 --sampler-mode 4 > gpurun_out/prof_r05sqm4.log 2>&1
bash tools/profile_train_traffic.sh r05train > gpurun_out/prof_r05train.log 2>&1
python bench.py > gpurun_out/bench_r05.json 2> gpurun_out/bench_r05.err
python bench.py --mode train > gpurun_out/bench_train_r05.json 2> gpurun_out/bench_train_r05.err
python bench.py --sampler-mode 4 --no-extras > gpurun_out/bench_r05_mode4.json 2> gpurun_out/bench_r05_mode4.err
python tools/bench_yaml_shapes.py > gpurun_out/yaml_shapes_r05.jsonl 2> gpurun_out/yaml_shapes_r05.err
python tools/bench_latency.py > gpurun_out/latency_r05.txt 2>&1
tail -2 gpurun_out/latency_r05.txt
