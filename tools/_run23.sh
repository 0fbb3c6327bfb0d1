bash tools/round_profiles.sh gpurun_out/r04final3 > /dev/null 2>&1
bash tools/secondary_measurements.sh gpurun_out/r04final3/secondary > /dev/null 2>&1
ls gpurun_out/r04final3 gpurun_out/r04final3/secondary | head -60
