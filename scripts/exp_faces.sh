run() { echo "lib $1 threads $2 grid $3"; SURTR_FACES_THREADS=$2 SURTR_FACES_WG=$3 python scripts/bench_flags.py $1 2>&1 | grep -E "flags (2|3)" | sed -e "s/'clip_pairs.*'refit'/'refit'/" -e "s/'out_scan.*//"; }
run surtr_amd/libsurtr_hip.so 256 1024
run surtr_amd/libsurtr_hip.so 128 1024
run surtr_amd/libsurtr_hip.so 64 1024
run build_tmp/libsurtr_hip_fl.so 256 1024
run build_tmp/libsurtr_hip_fl.so 128 1792
run build_tmp/libsurtr_hip_fl.so 64 1792
