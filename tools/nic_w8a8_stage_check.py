"""Stage-by-stage W8A8 comparison of one Swin block of the toy NIC (product on the GPU vs oracle on the CPU) from identical inputs:
shows which activation-quantisation points are exact and how many elements flip by one 8-bit level (see tests/test_gpu_nic.py)."""
import sys, os, numpy as np, torch
R=os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0,R); sys.path.insert(0,R+'/tests'); sys.path.insert(0,R+'/rdo-ptq_amd')
import test_gpu_nic as G
import test_oracle_golden as TG
from oracle import swin_oracle as S, rdo_oracle as O
from quantization import BaseQuantBlock, QuantModel, QuantModule
from quantization.quantizer import ActQuantizer
from helpers import WQ, AQ, T
from hipops import ops
gd=R+'/tests/golden'
fx, model = G.build(gd)
qnn = QuantModel(model=model, weight_quant_params=WQ, act_quant_params=AQ).cuda().eval()
cali = T(fx["cali"]).cuda()
qnn.set_quant_state(True, False)
with torch.no_grad(): qnn(cali[:2])
unit = qnn.model.g_a1
G.install_trained(fx, "g_a1", unit)
for m in unit.modules():
    if isinstance(m,(QuantModule,BaseQuantBlock)): m.trained=True
qnn.set_quant_state(True, True)
_, nic = TG._nic(gd); TG._install_calibrated_state(fx, nic)
st = nic.stages["g_a1"]; st.aq=True
x = T(fx["w8a8/g_a0"])
rel=lambda a,b: (float((a-b).abs().max()/(b.abs().max()+1e-12)), float(((a-b).abs()>1e-4*b.abs().max()).float().mean()))
with torch.no_grad():
    blk = unit.residual_group.blocks[0]
    pre="residual_group.blocks.0."
    B,C,H,W = x.shape
    t_o = x.flatten(2).transpose(1,2)
    t_g = x.cuda().permute(0,2,3,1).contiguous().view(B,H*W,C)
    n1_o = O.act_quant(st.ops[pre+"norm1"](t_o)); n1_g = blk.norm1(t_g)
    print("norm1", rel(n1_g.cpu(), n1_o))
    qkv_o = O.act_quant(st.ops[pre+"attn.qkv"](n1_o)); qkv_g = blk.attn.qkv(n1_o.cuda())
    print("qkv", rel(qkv_g.cpu(), qkv_o))
    raw_o = st.ops[pre+"attn.qkv"](n1_o); blk.attn.qkv.use_act_quant=False; raw_g = blk.attn.qkv(n1_o.cuda()); blk.attn.qkv.use_act_quant=True
    print("qkv raw", rel(raw_g.cpu(), raw_o))
    aq_g = ActQuantizer(raw_o.cuda()); print("AQ of same raw", rel(aq_g.cpu(), qkv_o))
    d=(aq_g.cpu()-qkv_o).abs(); i=d.argmax(); c=int(i)%raw_o.shape[-1]
    col=raw_o.reshape(-1,raw_o.shape[-1])[:,c]; print("worst ch", c, "min", float(col.min()), "max", float(col.max()), "diff", float(d.max()), "level", float((col-col.min()).abs().max()/255))
    # full attention from same input
    ws, shift = S.block_geometry(st.res, st.window_size, 0)
    xw = n1_o.view(B,H,W,C).view(B,H//ws,ws,W//ws,ws,C).permute(0,1,3,2,4,5).reshape(-1,ws*ws,C)
    a_o = S.window_attention(xw, st.ops[pre+"attn.qkv"], st.ops[pre+"attn.proj"], st.tables[0], st.heads, ws, None, True)
    a_o = a_o.view(B,H//ws,W//ws,ws,ws,C).permute(0,1,3,2,4,5).reshape(B,H*W,C)
    a_g = blk.attn.attend(n1_o.cuda(), B,H,W, blk.window_size, blk.shift_size)
    print("attend", rel(a_g.cpu(), a_o), blk.window_size, blk.shift_size, ws, shift)
    x1 = t_o + a_o
    n2_o = O.act_quant(st.ops[pre+"norm2"](x1)); n2_g = blk.norm2(x1.cuda()); print("norm2", rel(n2_g.cpu(), n2_o))
    m_o = torch.nn.functional.gelu(st.ops[pre+"mlp.fc1"](n2_o)); m_o = O.act_quant(m_o); m_o = O.act_quant(st.ops[pre+"mlp.fc2"](m_o))
    m_g = blk.mlp(n2_o.cuda()); print("mlp", rel(m_g.cpu(), m_o))
    f1_o = st.ops[pre+"mlp.fc1"](n2_o); f1_g = blk.mlp.fc1(n2_o.cuda()); print(" fc1", rel(f1_g.cpu(), f1_o))
    g_o = torch.nn.functional.gelu(f1_o); g_g = ops.gelu(f1_o.cuda().contiguous()); print(" gelu", rel(g_g.cpu(), g_o))
    ga_o = O.act_quant(g_o); ga_g = blk.mlp._aq(g_o.cuda()); print(" aq(gelu)", rel(ga_g.cpu(), ga_o), blk.mlp.use_act_quant, blk.mlp.trained)
    f2_o = O.act_quant(st.ops[pre+"mlp.fc2"](ga_o)); f2_g = blk.mlp.fc2(ga_o.cuda()); print(" fc2", rel(f2_g.cpu(), f2_o))
    b_o = O.act_quant(x1 + m_o); b_g = blk(t_g, (H,W)); 
    b_oo = S.swin_block(t_o,(H,W),st.ops,pre,st.tables[0],st.heads,ws,shift,True)
    print("block", rel(b_g.cpu(), b_oo), rel(b_o, b_oo))
