tools/valu_rate.sh > gpurun_out/valu_rate.log 2>&1
tools/collect_profiles.sh r05_KT KT > gpurun_out/r05_KT.log 2>&1
tools/collect_profiles.sh r05_SY SY > gpurun_out/r05_SY.log 2>&1
tools/collect_profiles.sh r05_NS NS > gpurun_out/r05_NS.log 2>&1
tail -3 gpurun_out/r05_KT.log
