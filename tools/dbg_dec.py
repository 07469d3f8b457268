import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/audio-style-transfer_amd')
import torch, ast_amd
from oracle import seeded_params as sp
from ast_amd import ops, layers as L
dec = ast_amd.Decoder()
dec.load_state_dict(sp.seeded_state_dict(dec.state_dict(), tag="decoder"))
dec = dec.cuda().train()
x = sp.seeded_input(2, 2).cuda()
y = x[..., :513]
dec._prepare()
h = ops.nchw_to_nhwc(y.view(4, 2, 287, 513), L.img_dtype())
tr = True
def watch(t, name):
    t.register_hook(lambda g: print(name, 'grad norm', float(g.norm()), tuple(g.shape)))
    return t
from ast_amd.new_decoder import ENC_CH
for pw, i, (_, _, s) in zip(dec._enc, (0, 3, 6, 9), ENC_CH):
    c = watch(L.conv(h, pw, 3, s, 1, False), f'conv{i}')
    h = watch(L.bn_act(c, dec.conv_encoder[i + 1], tr, relu=True), f'bn{i}')
h = watch(ops.AdaptivePoolFn.apply(h, 32, 16), 'pool')
c = watch(L.conv(h, dec._sp[0], 3, 1, 1, False), 'sp0')
h = watch(L.bn_act(c, dec.spatial_projection[1], tr, relu=True), 'spbn')
h = watch(L.conv(h, dec._sp[1], 1, 1, 0, True), 'sp3')
sl = watch(h[..., 0].reshape(h.shape[0], -1), 'slice')
flat = watch(ops.CastFn.apply(sl, torch.float32), 'cast')
emb = L.linear(flat, dec._f2s)
emb.sum().backward()
