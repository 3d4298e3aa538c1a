import sys; sys.path.insert(0,'.')
order = sys.argv[1]
if order == 'torch_first':
    import torch; print('torch cuda', torch.cuda.is_available(), torch.cuda.device_count(), flush=True)
    from ldso_amd import binding; print('mylib devices', binding.lib().ldso_device_count(), flush=True)
    x = torch.zeros(4, device='cuda'); print(x.sum().item())
else:
    from ldso_amd import binding; print('mylib devices', binding.lib().ldso_device_count(), flush=True)
    import torch; print('torch cuda', torch.cuda.is_available(), flush=True)
