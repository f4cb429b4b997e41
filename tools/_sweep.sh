run() { echo "$1: $(eval "$2 timeout 300 python bench.py --steps 200 --warmup 8 --no-train-object --no-cpu-baseline $3" 2>gpurun_out/sw.err > gpurun_out/sw.json; python tools/bench_line.py gpurun_out/sw.json)"; }
run "base t0" "RA_ENGINE_TICKETS=0" ""
run "fuse-patch-pairs t0" "RA_ENGINE_TICKETS=0" "--fuse-patch-pairs"
run "fuse-patch-pairs t1" "RA_ENGINE_TICKETS=1" "--fuse-patch-pairs"
run "fuse extract conv0 t0" "RA_ENGINE_TICKETS=0 RA_FUSE_EXTRACT_CONV0=1" ""
run "coalesce 4, in-flight 16, t0" "RA_ENGINE_TICKETS=0" "--coalesce 4 --in-flight 16"
run "coalesce 4, in-flight 16, t1" "RA_ENGINE_TICKETS=1" "--coalesce 4 --in-flight 16"
run "coalesce 3, in-flight 12, t1" "RA_ENGINE_TICKETS=1" "--coalesce 3 --in-flight 12"
run "coalesce 2, in-flight 8, 3 streams t1" "RA_ENGINE_TICKETS=1" "--streams 3"
