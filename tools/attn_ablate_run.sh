# diagnostic build + ablation table of the attention kernels (on the GPU box; the tree there is scratch)
cd rdo-ptq_amd/csrc && make clean >/dev/null && make -j16 DIAG=1 >/dev/null 2>&1; cd ../..
python tools/attn_ablate.py 2>&1 | grep -v amdgpu.ids
