cd rdo-ptq_amd/csrc && make clean >/dev/null && make -j16 DIAG=1 >/dev/null 2>&1; cd ../..
for o in 1 2 3; do echo "== max occupancy $o"; RDO_ATTN_MAXOCC=$o python tools/attn_ablate.py 2>&1 | grep -v amdgpu.ids | head -6; done
