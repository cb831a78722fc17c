"""CPU restatement (plain PyTorch ops, differentiable) of the host-side stages around the
rasterizer: mesh binding and the photometric loss.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).

Follows, op for op, the reference's PyTorch code:
  compute_face_orientation      utils/graphics_utils.py:92-137   (pinned by tests/golden/face_orientation.npz)
  update_face_coor              scene/mesh_gaussian_model.py:90-95
  get_scaling/get_rotation/get_xyz  scene/mesh_gaussian_model.py:105-128
  get_barycentric_3d / get_xyz  scene/avatar_gaussian_model.py:140-159
  l1_loss / ssim                utils/loss_utils.py:17-68        (pinned by tests/golden/loss.npz)
roma (absent from this image) is restated from its documented algorithm:
  rotmat_to_unitquat = branch on argmax(diag, trace) (the SciPy construction), xyzw output;
  quat_product = Hamilton product.  These two are "parity unpinned" against roma itself.
"""
import torch
import torch.nn.functional as F


def _dot(x, y):
    return torch.sum(x * y, -1, keepdim=True)


def _length(x, eps=1e-20):
    return torch.sqrt(torch.clamp(_dot(x, x), min=eps))


def _safe_normalize(x, eps=1e-20):
    return x / _length(x, eps)


def compute_face_orientation(verts, faces):
    i0, i1, i2 = faces[..., 0].long(), faces[..., 1].long(), faces[..., 2].long()
    v0, v1, v2 = verts[..., i0, :], verts[..., i1, :], verts[..., i2, :]
    a0 = _safe_normalize(v1 - v0)
    a1 = _safe_normalize(torch.cross(a0, v2 - v0, dim=-1))
    a2 = -_safe_normalize(torch.cross(a1, a0, dim=-1))
    orientation = torch.cat([a0[..., None], a1[..., None], a2[..., None]], dim=-1)
    scale = (_length(v1 - v0) + _dot(a2, (v2 - v0)).abs()) / 2
    return orientation, scale


def rotmat_to_unitquat_xyzw(R):
    """roma.rotmat_to_unitquat restated (SciPy's from_matrix construction)."""
    m = R.reshape(-1, 3, 3)
    n = m.shape[0]
    dec = torch.empty(n, 4, dtype=m.dtype)
    dec[:, :3] = m.diagonal(dim1=1, dim2=2)
    dec[:, 3] = dec[:, :3].sum(1)
    choice = dec.argmax(1)
    out = []
    for b in range(4):
        sel = torch.nonzero(choice == b).reshape(-1)
        mb, db = m[sel], dec[sel]
        if b < 3:
            i, j, k = b, (b + 1) % 3, (b + 2) % 3
            comp = [None] * 4
            comp[i] = 1 - db[:, 3] + 2 * mb[:, i, i]
            comp[j] = mb[:, j, i] + mb[:, i, j]
            comp[k] = mb[:, k, i] + mb[:, i, k]
            comp[3] = mb[:, k, j] - mb[:, j, k]
        else:
            comp = [mb[:, 2, 1] - mb[:, 1, 2], mb[:, 0, 2] - mb[:, 2, 0], mb[:, 1, 0] - mb[:, 0, 1], 1 + db[:, 3]]
        out.append((sel, torch.stack(comp, -1)))
    q = torch.zeros(n, 4, dtype=m.dtype)
    for sel, val in out:
        q = q.index_put((sel,), val)
    return q / torch.norm(q, dim=1, keepdim=True)


def quat_product_wxyz(a, b):
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack([aw * bw - ax * bx - ay * by - az * bz,
                        aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx,
                        aw * bz + ax * by - ay * bx + az * bw], -1)


def mesh_bind(verts, faces, binding, local_xyz, log_scaling, raw_rot, bary=None):
    """-> xyz [P,3], scaling [P,3], rotation [P,4] exactly as the reference's getters compose them."""
    face_center = verts[faces].mean(1)
    R, s = compute_face_orientation(verts, faces)
    q = rotmat_to_unitquat_xyzw(R)
    face_quat = torch.cat([q[:, 3:], q[:, :3]], -1)                       # quat_xyzw_to_wxyz
    scaling = torch.exp(log_scaling) * s[binding]
    rot = F.normalize(raw_rot)
    fq = F.normalize(face_quat[binding])
    rotation = F.normalize(quat_product_wxyz(fq, rot))
    xyz = torch.bmm(R[binding], local_xyz[..., None]).squeeze(-1) * s[binding]
    if bary is None:
        xyz = xyz + face_center[binding]
    else:
        tri = verts[faces][binding]
        xyz = xyz + bary[:, 0:1] * tri[:, 0] + bary[:, 1:2] * tri[:, 1] + bary[:, 2:3] * tri[:, 2]
    return xyz, scaling, rotation


# ---- photometric loss (utils/loss_utils.py) -------------------------------------------------
def l1_loss(out, gt, mask=None):
    return torch.abs(out - gt).mean() if mask is None else torch.abs((out - gt) * mask).mean()


def _window(ws, ch, dtype):
    g = torch.tensor([__import__("math").exp(-(x - ws // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(ws)])
    g = (g / g.sum()).unsqueeze(1)
    return g.mm(g.t()).float().unsqueeze(0).unsqueeze(0).expand(ch, 1, ws, ws).contiguous().to(dtype)


def ssim(img1, img2, mask=None, window_size=11):
    """NB the reference multiplies img1/img2 by the mask IN PLACE (loss_utils.py:44-46); here out of place."""
    ch = img1.size(-3)
    w = _window(window_size, ch, img1.dtype)
    if mask is not None:
        img1, img2 = img1 * mask, img2 * mask
    p = window_size // 2
    mu1, mu2 = F.conv2d(img1, w, padding=p, groups=ch), F.conv2d(img2, w, padding=p, groups=ch)
    mu1_sq, mu2_sq, mu12 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = F.conv2d(img1 * img1, w, padding=p, groups=ch) - mu1_sq
    s2 = F.conv2d(img2 * img2, w, padding=p, groups=ch) - mu2_sq
    s12 = F.conv2d(img1 * img2, w, padding=p, groups=ch) - mu12
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))).mean()
