# Round 6: scripts/aser_drift_probe.py -- what grows with the step count in the ASER leg
T=${1:-r6ao}
mkdir -p gpurun_out
timeout -k 10 600 python scripts/aser_drift_probe.py --repeats 12 > gpurun_out/${T}_aser_drift.txt 2> gpurun_out/${T}_aser_drift.err; echo rc=$?
cat gpurun_out/${T}_aser_drift.txt | cut -c1-260; tail -3 gpurun_out/${T}_aser_drift.err
