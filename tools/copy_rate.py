import torch, json
n = 4096 * 49 * 64
a = torch.randn(n, device="cuda"); b = torch.empty_like(a)
def t(f, reps=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
mb = n * 4 / 1e6
us_copy = t(lambda: b.copy_(a))
us_fill = t(lambda: b.fill_(1.0))
us_sum = t(lambda: a.sum())
c = torch.empty_like(a)
us_add = t(lambda: torch.add(a, b, out=c))
print(json.dumps({"tensor_MB": mb, "copy_us": us_copy, "copy_TBps": 2 * mb / us_copy, "fill_us": us_fill, "fill_TBps": mb / us_fill,
                  "sum_us": us_sum, "sum_TBps": mb / us_sum, "add_us": us_add, "add_TBps": 3 * mb / us_add}))
