import sys
import torch
g0 = torch.load(sys.argv[1]); g2 = torch.load(sys.argv[2])
for n in ['model.lisa_iou_head.0.weight', 'model.lisa_iou_head.0.bias', 'model.lisa_embedding_head.0.bias', 'model.lisa_iou_head.2.weight', 'model.lisa_iou_head.2.bias']:
    a, b = g0[n], g2[n]; d = (a - b).abs()
    print(n, tuple(a.shape), 'max', d.max().item(), 'n>1e-3', int((d > 1e-3).sum()), 'of', d.numel())
    if d.dim() == 2:
        r = d.max(1).values; print('  rows with diff>1e-3:', int((r > 1e-3).sum()), 'top rows', r.topk(5))
    else:
        print('  top', d.topk(5))
