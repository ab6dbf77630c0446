"""generate_anchors (reference yolov6/assigners/anchor_generator.py:6-63): host-side grid
construction, a few KB per call; the eval-mode grid is generated inside the head-decode
kernel instead (yolov6_amd/csrc/head_decode.hip)."""
import torch


def generate_anchors(feats, fpn_strides, grid_cell_size=5.0, grid_cell_offset=0.5, device='cpu', is_eval=False,
                     mode='af'):
    assert feats is not None
    rep = 3 if mode == 'ab' else 1
    points, strides, boxes, counts = [], [], [], []
    for feat, stride in zip(feats, fpn_strides):
        h, w = feat.shape[-2:]
        scale = 1 if is_eval else stride
        sx = (torch.arange(end=w, device=device) + grid_cell_offset) * scale
        sy = (torch.arange(end=h, device=device) + grid_cell_offset) * scale
        gy, gx = torch.meshgrid(sy, sx, indexing='ij')
        dt = torch.float if is_eval else feats[0].dtype
        pt = torch.stack([gx, gy], -1).to(dt).reshape(-1, 2).repeat(rep, 1)
        points.append(pt)
        strides.append(torch.full((pt.shape[0], 1), stride, dtype=dt, device=device if is_eval else None))
        if not is_eval:
            half = grid_cell_size * stride * 0.5
            bx = torch.stack([gx - half, gy - half, gx + half, gy + half], -1).to(dt).reshape(-1, 4).repeat(rep, 1)
            boxes.append(bx)
            counts.append(bx.shape[0])
    if is_eval:
        return torch.cat(points), torch.cat(strides)
    return torch.cat(boxes), torch.cat(points).to(device), counts, torch.cat(strides).to(device)
