import torch, torch.nn as nn
torch.manual_seed(0)
layer = nn.TransformerEncoderLayer(32, 2, 128, 0.1, "gelu", batch_first=True, norm_first=True).eval()
x = torch.randn(2, 17, 32)
with torch.no_grad():
    ref = layer(x.double() if False else x)
    refd = nn.TransformerEncoderLayer(32, 2, 128, 0.1, "gelu", batch_first=True, norm_first=True).double().eval()
    refd.load_state_dict({k: v.double() for k, v in layer.state_dict().items()})
    r64 = refd(x.double())
    g = layer.cuda()
    for fast in (True, False):
        torch.backends.mha.set_fastpath_enabled(fast)
        o = g(x.cuda()).cpu()
        print("fastpath", fast, "gpu-vs-f64", float((o.double()-r64).abs().max()), "cpu-vs-f64", float((ref.double()-r64).abs().max()))
    # plain matmul check
    a, b = torch.randn(64, 64), torch.randn(64, 64)
    print("matmul gpu-vs-f64", float(((a.cuda()@b.cuda()).cpu().double() - a.double()@b.double()).abs().max()), "cpu", float(((a@b).double() - a.double()@b.double()).abs().max()))
    q = torch.randn(2,2,17,16)
    import torch.nn.functional as F
    o = F.scaled_dot_product_attention(q.cuda(), q.cuda(), q.cuda()).cpu()
    o64 = F.scaled_dot_product_attention(q.double(), q.double(), q.double())
    print("sdpa gpu-vs-f64", float((o.double()-o64).abs().max()))
    w = torch.randn(96, 32); bb = torch.randn(96)
    print("linear gpu", float((F.linear(x.cuda(), w.cuda(), bb.cuda()).cpu().double() - F.linear(x.double(), w.double(), bb.double())).abs().max()))
    print("gelu gpu", float((F.gelu(x.cuda()).cpu().double()-F.gelu(x.double())).abs().max()), "ln", float((F.layer_norm(x.cuda(),(32,)).cpu().double()-F.layer_norm(x.double(),(32,))).abs().max()))
