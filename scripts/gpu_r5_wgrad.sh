# Round 5, weight gradient (~40 s of box time, no torch): where a pixel tile's time goes (kbench wgradtrace: s_memtime stamps at the phase
# boundaries, measurement builds of the hot forms), and the two schedule changes written blind at the end of round 4 (OCL_WGRAD_XCD=1: XCD-aware workgroup order; OCL_WGRAD_PD=2: prefetch distance of two tiles) --
# per layer against the reference kernel, the phase timeline under it, and bit for bit through the whole pass.
# gpurun --timeout 200 -- 'bash scripts/gpu_r5_wgrad.sh r5w'
mkdir -p gpurun_out
T=${1:-r5w}
cd online-continual-learning_amd/csrc
O=../../gpurun_out/${T}_wgrad.txt
{
  for E in "OCL_NONE=1" "OCL_WGRAD_XCD=1" "OCL_WGRAD_Q=1"; do
    echo "### [$E] kbench 220 2 32 wgradtrace"; env $E timeout 60 ./kbench 220 2 32 wgradtrace 2>&1 | grep -v "^#"
  done
  echo "### [default] kbench 20 1 32 wgradtrace"; timeout 60 ./kbench 20 1 32 wgradtrace 2>&1 | grep -v "^#"
  for E in "OCL_NONE=1" "OCL_WGRAD_XCD=1" "OCL_WGRAD_PD=2" "OCL_WGRAD_XCD=1 OCL_WGRAD_PD=2"; do
    echo "### [$E] kbench 220 2 32 wgrad"; env $E timeout 60 ./kbench 220 2 32 wgrad 2>&1 | grep -E "wgrad |MISMATCH|rror" | cut -c1-20,96-240
  done
  for cfg in "220 2 32 1" "20 1 32 0" "20 1 84 0"; do
    echo "### netcheck $cfg: default -> file; OCL_WGRAD_XCD=1 / OCL_WGRAD_PD=2 compared (order-independent sums: must be bit-identical)"
    OCL_DETERMINISTIC=1 timeout 60 ./netcheck $cfg write /tmp/ref.bin | head -1
    OCL_DETERMINISTIC=1 OCL_WGRAD_XCD=1 timeout 60 ./netcheck $cfg compare /tmp/ref.bin
    OCL_DETERMINISTIC=1 OCL_WGRAD_PD=2 timeout 60 ./netcheck $cfg compare /tmp/ref.bin
    echo "# pass time, default sums: default / XCD / PD=2 / both"
    timeout 60 ./netcheck $cfg write /tmp/ref2.bin | head -1
    for E in "OCL_WGRAD_XCD=1" "OCL_WGRAD_PD=2" "OCL_WGRAD_XCD=1 OCL_WGRAD_PD=2" "OCL_REDUCE_GROUP=4" "OCL_REDUCE_GROUP=2"; do echo -n "[$E] "; env $E timeout 60 ./netcheck $cfg compare /tmp/ref2.bin | grep -E "netcheck|beyond" | tr '\n' ' '; echo; done
    echo "# OCL_REDUCE_GROUP=4, order-independent sums: must be bit-identical"
    OCL_DETERMINISTIC=1 OCL_REDUCE_GROUP=4 timeout 60 ./netcheck $cfg compare /tmp/ref.bin | tail -1
  done
  echo "### conv_q_kernel<2,12,*> (layer 1 at 220 views): phase timeline, KBENCH_TRACE=1 kbench 220 2 32 conv 0"
  KBENCH_TRACE=1 timeout 90 ./kbench 220 2 32 conv 0 2>&1 | grep -E "^conv1|^layer1|trace conv_q|conv_q "
} > $O 2>&1
cut -c1-260 $O
