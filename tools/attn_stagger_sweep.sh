# stagger sweep of the attention kernels: value = 64-cycle units per residency slot, + 65536 x (workgroups per CU - 1) to key on consecutive ids
for s in 0 $((65536+100)) $((65536+200)) $((131072+70)) $((131072+140)); do echo "== stagger $s"; RDO_ATTN_STAGGER_FWD=$s RDO_ATTN_STAGGER_BWD=$s python tools/attn_ablate.py 2>&1 | grep -v amdgpu.ids | cut -c1-45 | head -6; done
