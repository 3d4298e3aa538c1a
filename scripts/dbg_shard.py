import sys, copy
import numpy as np, torch
sys.path.insert(0, '.')
from ldso_amd import synth, binding
win = synth.make_config('small')
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); st = ts.cuda_stream
ref = binding.BA.from_window(win, stream=st)
ref.collect_active(); ref.linearize_all(False); ref.apply_res()
refbuf = torch.zeros(ref.gn_reduce_doubles(), dtype=torch.float64, device='cuda')
ref.gn_reduce_local(refbuf.data_ptr(), 1e-1); torch.cuda.synchronize()
half = win.P // 2
ranks, bufs = [], []
for (a, b) in ((0, half), (half, win.P)):
    g = binding.BA.from_window(win, stream=st)
    g.set_shard(a, b)
    g.collect_active(); g.linearize_all(False); g.apply_res()
    ranks.append(g); bufs.append(torch.zeros(g.gn_reduce_doubles(), dtype=torch.float64, device='cuda'))
for g, b in zip(ranks, bufs): g.gn_reduce_local(b.data_ptr(), 1e-1)
torch.cuda.synchronize()
tot = (bufs[0] + bufs[1]).cpu().numpy(); r = refbuf.cpu().numpy()
n = 8 * win.F + 4; blk = n * n + n
print('H lower maxabs diff', np.abs(tot[:n*n] - r[:n*n]).max(), 'of', np.abs(r[:n*n]).max())
print('b diff', np.abs(tot[n*n:blk] - r[n*n:blk]).max(), 'of', np.abs(r[n*n:blk]).max())
print('scalars', tot[blk:blk+8], r[blk:blk+8])
print('cand equal', np.array_equal(tot[blk+8:], r[blk+8:]))
for i, b in enumerate(bufs):
    x = b.cpu().numpy(); print('rank', i, 'H maxabs', np.abs(x[:n*n]).max(), 'b maxabs', np.abs(x[n*n:blk]).max())
# full iterations
ref2 = binding.BA.from_window(win, stream=st)
ref2.collect_active(); ref2.linearize_all(False); ref2.apply_res()
for it in range(3):
    ref2.enqueue_gn(it, 1); ref2.sync()
    if it > 0:
        for g, b in zip(ranks, bufs): g.gn_reduce_local(b.data_ptr(), 1e-1)
    tot = bufs[0] + bufs[1]
    for g, b in zip(ranks, bufs):
        b.copy_(tot); g.gn_solve_reduced(b.data_ptr(), it, 1e-1)
    torch.cuda.synchronize()
    fr = ref2.get_frames()['frames']['state']
    for i, g in enumerate(ranks):
        fg = g.get_frames()
        print('it', it, 'rank', i, 'state maxdiff', np.abs(fg['frames']['state'] - fr).max(), 'of', np.abs(fr).max(), 'TH', fg['frames']['frameEnergyTH'][-1], ref2.get_frames()['frames']['frameEnergyTH'][-1])
print('---- detailed')
ref3 = binding.BA.from_window(win, stream=st)
ref3.collect_active(); ref3.linearize_all(False); ref3.apply_res()
ranks, bufs = [], []
for (a, b) in ((0, half), (half, win.P)):
    g = binding.BA.from_window(win, stream=st); g.set_shard(a, b)
    g.collect_active(); g.linearize_all(False); g.apply_res()
    ranks.append(g); bufs.append(torch.zeros(g.gn_reduce_doubles(), dtype=torch.float64, device='cuda'))
rb = torch.zeros(ref3.gn_reduce_doubles(), dtype=torch.float64, device='cuda')
for it in range(4):
    ref3.gn_reduce_local(rb.data_ptr(), 1e-1)
    for g, b in zip(ranks, bufs): g.gn_reduce_local(b.data_ptr(), 1e-1)
    torch.cuda.synchronize()
    tot = bufs[0] + bufs[1]
    t = tot.cpu().numpy(); r = rb.cpu().numpy()
    c_t = t[blk+8:]; c_r = r[blk+8:]
    print('it', it, 'E', t[blk], r[blk], 'ncand', (c_t > 0).sum(), (c_r > 0).sum(), 'cand maxdiff', np.abs(c_t - c_r).max(), 'H diff', np.abs(t[:n*n]-r[:n*n]).max(), 'b diff', np.abs(t[n*n:blk]-r[n*n:blk]).max())
    ref3.gn_solve_reduced(rb.data_ptr(), it, 1e-1)
    for g, b in zip(ranks, bufs):
        b.copy_(tot); g.gn_solve_reduced(b.data_ptr(), it, 1e-1)
    torch.cuda.synchronize()
