run() { echo "$1: $(eval "$2 timeout 300 python bench.py --steps 240 --warmup 8 --no-train-object --no-cpu-baseline $3" 2>gpurun_out/sw.err > gpurun_out/sw.json; python tools/bench_line.py gpurun_out/sw.json 2>/dev/null || tail -2 gpurun_out/sw.err)"; }
python -c "import torch; print(torch.cuda.Stream.priority_range())"
run "base" "" ""
run "prio -1,0,0,0" "RA_PIPE_PRIO=-1,0,0,0" ""
run "prio -1,-1,0,0" "RA_PIPE_PRIO=-1,-1,0,0" ""
run "prio -1,0,-1,0" "RA_PIPE_PRIO=-1,0,-1,0" ""
run "prio -1 all" "RA_PIPE_PRIO=-1" ""
run "prio -1,0,0,0 tickets" "RA_PIPE_PRIO=-1,0,0,0 RA_ENGINE_TICKETS=1" ""
