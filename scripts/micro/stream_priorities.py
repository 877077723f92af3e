import torch
for p in (-3,-2,-1,0,1,2):
    try:
        s = torch.cuda.Stream(priority=p); print(p, "->", s.priority)
    except Exception as e:
        print(p, "err", repr(e)[:100])
print(torch.cuda.current_stream().priority)
