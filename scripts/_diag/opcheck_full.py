import sys, torch
sys.path.insert(0, '.')
from news_recommendation_model_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
B,T,H,D = 2,3,5,16
t, h = r(B,T,D).requires_grad_(), r(B,H,D).requires_grad_()
w1, b1 = (0.1*r(D,4*D)).requires_grad_(), (0.1*r(D)).requires_grad_()
w2, b2 = r(1,D).requires_grad_(), r(1).requires_grad_()
for name, args in [("pwattn_fwd", (t,h,w1,b1,w2,b2,True,0)), ("weighted_pool_fwd", (r(2,3,5).requires_grad_(), h)),
                   ("linear_fwd", (r(37,24).requires_grad_(), r(10,24).requires_grad_(), r(10).requires_grad_(), True)),
                   ("mlp_gelu_fwd", (r(37,24).requires_grad_(), r(10,24).requires_grad_(), r(10).requires_grad_(), r(24,10).requires_grad_(), r(24).requires_grad_(), r(37,24).requires_grad_()))]:
    try:
        res = torch.library.opcheck(getattr(torch.ops.nrm, name).default, args)
        print(name, "OK", res)
    except Exception as e:
        print(name, "FAIL", type(e).__name__, str(e)[:600])
