tools/collect_profiles.sh r06_KT KT > gpurun_out/r06_KT.log 2>&1
tools/collect_profiles.sh r06_SY SY > gpurun_out/r06_SY.log 2>&1
tools/collect_profiles.sh r06_NS NS > gpurun_out/r06_NS.log 2>&1
tail -3 gpurun_out/r06_KT.log
