#!/bin/bash
# Copies what a full measurement run (tools/round_profiles.sh + tools/secondary_measurements.sh + tools/lr_driver_profile.sh, as
# driven by the round's scratch script) left under gpurun_out/<dir> into profiles/ under this round's names.
#   tools/collect_round_profiles.sh gpurun_out/r05final r05
set -e
src=$1; r=$2; p=profiles
cp $src/round/bench.json $p/${r}_bench.json
cp $src/round/bench_C2.json $p/${r}_bench_C2_n8192.json
cp $src/round/bench_C4.json $p/${r}_bench_C4_L8.json
cp $src/round/bench_C5.json $p/${r}_bench_C5_n32768.json
cp $src/round/kt_C3/kt_kernel_stats.csv $p/${r}_bench_kernel_stats.csv
cp $src/round/kts_C3/kts_kernel_stats.csv $p/${r}_bench_kernel_stats_serial_chunks.csv
cp $src/round/kt_C2/kt_kernel_stats.csv $p/${r}_bench_C2_kernel_stats.csv
cp $src/round/kts_C2/kts_kernel_stats.csv $p/${r}_bench_C2_kernel_stats_serial_chunks.csv
cp $src/round/pmc_traffic_C3.json $p/${r}_bench_pmc_traffic.json
cp $src/round/pmc_traffic_C2.json $p/${r}_bench_C2_pmc_traffic.json
cp $src/round/sq_counters_C3.json $p/${r}_bench_sq_counters.json
cp $src/round/sq_counters_C2.json $p/${r}_bench_C2_sq_counters.json
cp $src/round/batch_sweep_C3.json $p/${r}_batch_sweep_C3.json
cp $src/round/kt_lt_direct/kt_kernel_stats.csv $p/${r}_lt_direct_d512_kernel_stats.csv
cp $src/round/lt_direct_probe.txt $p/${r}_lt_direct_d512_probe.txt
cp $src/round/lt_direct_timeline.txt $p/${r}_lt_direct_d512_timeline.txt
mkdir -p $p/$r/secondary $p/$r/lr_driver_2000_paused
cp -r $src/secondary/* $p/$r/secondary/
cp $src/lr_driver_2000/run*.txt $p/$r/lr_driver_2000_paused/
cp $src/lr_driver.log $p/$r/lr_driver_2000_paused/wall_times_6_runs_2s_pause.txt
cp $src/stress_parity.txt $p/$r/stress_parity.txt
cp $src/pytest.log $p/$r/gpu_suite.log
