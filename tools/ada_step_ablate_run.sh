# diagnostic build + ablation table of the AdaRound step (on the GPU box; the tree there is scratch)
mkdir -p gpurun_out
cd rdo-ptq_amd/csrc && make clean >/dev/null && make -j16 DIAG=1 >/dev/null 2>&1; cd ../..
python tools/ada_step_ablate.py --json gpurun_out/ada_step_ablate.json 2>&1 | grep -v amdgpu.ids
