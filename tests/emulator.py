"""Reference semantics of every entry point in include/tartangan_amd.h, written
with plain PyTorch ops.  TEST INFRASTRUCTURE ONLY (lives under tests/).

Two uses:
  * ``-m "not gpu"``: installed through ``backend._set_backend_for_testing`` so the
    host logic (autograd composition, double backward, trainers, optimiser,
    data-parallel glue) can be checked on CPU against the oracle and the golden
    fixtures.  The product never falls back to this.
  * ``-m gpu``: per-kernel expected values for the HIP kernels (same call, CPU
    copies of the same inputs).
"""
import math

import torch
import torch.nn.functional as F


def _v(t, *shape):
    return t.view(*shape)


class Emulator:
    name = 'emulator'

    # ---------------------------------------------------------------- conv
    def conv2d_fwd(self, x, w, bias, residual, y, B, Cin, Cout, H, W, ks):
        r = F.conv2d(_v(x, B, Cin, H, W), _v(w, Cout, Cin, ks, ks), bias, padding=ks // 2)
        y.copy_(r if residual is None else _v(residual, B, Cout, H, W) + r)
        return 0

    def conv2d_fwd_up2res(self, x, w, bias, residual_lo, y, B, Cin, Cout, H, W):
        r = F.conv2d(_v(x, B, Cin, H, W), _v(w, Cout, Cin, 3, 3), bias, padding=1)
        y.copy_(F.interpolate(_v(residual_lo, B, Cout, H // 2, W // 2), scale_factor=2) + r)
        return 0

    def upconv3x3_weights(self, w, wp, Cout, Cin):
        k = _v(w, Cout, Cin, 3, 3)
        out = _v(wp, 4, Cout, Cin, 2, 2)
        for dy in range(2):
            rows = [k[:, :, 0], k[:, :, 1] + k[:, :, 2]] if dy == 0 else [k[:, :, 0] + k[:, :, 1], k[:, :, 2]]
            for dx in range(2):
                for ty in range(2):
                    r = rows[ty]
                    cols = [r[:, :, 0], r[:, :, 1] + r[:, :, 2]] if dx == 0 else [r[:, :, 0] + r[:, :, 1], r[:, :, 2]]
                    out[dy * 2 + dx, :, :, ty, 0] = cols[0]
                    out[dy * 2 + dx, :, :, ty, 1] = cols[1]
        return 0

    def upconv3x3_fwd(self, a, wp, bias, residual, y, B, Cin, Cout, H, W):
        ap = F.pad(_v(a, B, Cin, H, W), (1, 1, 1, 1))
        out = _v(y, B, Cout, 2 * H, 2 * W)
        res = None if residual is None else _v(residual, B, Cout, 2 * H, 2 * W)
        k = _v(wp, 4, Cout, Cin, 2, 2)
        for dy in range(2):
            for dx in range(2):
                full = F.conv2d(ap, k[dy * 2 + dx], bias)             # (H+1) x (W+1): window starting at row i'-1
                r = full[:, :, dy:dy + H, dx:dx + W]
                out[:, :, dy::2, dx::2] = r if res is None else res[:, :, dy::2, dx::2] + r
        return 0

    def upconv3x3_weights_t(self, w, w4t, Cout, Cin):
        k = _v(w, Cout, Cin, 3, 3)
        sets = [[2], [1, 2], [0, 1], [0]]
        out = _v(w4t, Cin, Cout, 4, 4)
        for u in range(4):
            for v in range(4):
                acc = 0
                for kh in sets[u]:
                    for kw in sets[v]:
                        acc = acc + k[:, :, kh, kw]
                out[:, :, u, v] = acc.t()
        return 0

    def upconv3x3_weights_pair(self, w, wp, w4t, Cout, Cin):
        self.upconv3x3_weights(w, wp, Cout, Cin)
        return self.upconv3x3_weights_t(w, w4t, Cout, Cin)

    def upconv3x3_dgrad_supported(self, B, Cin, Cout, H, W):
        return 1

    def upconv3x3_dgrad(self, gy, w4t, ga, B, Cin, Cout, H, W):
        _v(ga, B, Cin, H, W).copy_(F.conv2d(_v(gy, B, Cout, 2 * H, 2 * W), _v(w4t, Cin, Cout, 4, 4), stride=2, padding=1))
        return 0

    def poolconv3x3_weights(self, w, w4, wp, Cout, Cin):
        k = _v(w, Cout, Cin, 3, 3) * 0.25
        full = F.conv2d(F.pad(k.reshape(-1, 1, 3, 3), (1, 1, 1, 1)), torch.ones(1, 1, 2, 2))       # sum over (dy, dx)
        _v(w4, Cout, Cin, 4, 4).copy_(full.view(Cout, Cin, 4, 4))
        out = _v(wp, 4, Cin, Cout, 2, 2)
        for dy in range(2):
            rows = [k[:, :, 2], k[:, :, 1] + k[:, :, 0]] if dy == 0 else [k[:, :, 2] + k[:, :, 1], k[:, :, 0]]
            for dx in range(2):
                for ty in range(2):
                    r = rows[ty]
                    cols = [r[:, :, 2], r[:, :, 1] + r[:, :, 0]] if dx == 0 else [r[:, :, 2] + r[:, :, 1], r[:, :, 0]]
                    out[dy * 2 + dx, :, :, ty, 0] = cols[0].t()
                    out[dy * 2 + dx, :, :, ty, 1] = cols[1].t()
        return 0

    def poolconv3x3_supported(self, B, Cin, Cout, H, W):
        return 1

    def poolconv3x3_fwd(self, x, w4, bias, residual, y, B, Cin, Cout, H, W):
        r = F.conv2d(_v(x, B, Cin, 2 * H, 2 * W), _v(w4, Cout, Cin, 4, 4), bias, stride=2, padding=1)
        _v(y, B, Cout, H, W).copy_(r if residual is None else _v(residual, B, Cout, H, W) + r)
        return 0

    def poolconv3x3_dgrad(self, gy, wp, gx, B, Cin, Cout, H, W):
        return self.upconv3x3_fwd(gy, wp, None, None, gx, B, Cout, Cin, H, W)

    def poolconv3x3_wgrad_workspace(self, B, Cin, Cout, H, W):
        return (Cout * Cin * 9 + Cout) * 4

    def poolconv3x3_wgrad(self, x, gy, gw, ws, ws_bytes, B, Cin, Cout, H, W, accumulate, gbias=None):
        g_hi = (_v(gy, B, Cout, H, W) * 0.25).repeat_interleave(2, 2).repeat_interleave(2, 3)
        r = torch.nn.grad.conv2d_weight(_v(x, B, Cin, 2 * H, 2 * W), (Cout, Cin, 3, 3), g_hi, padding=1)
        gw.copy_(gw + r if accumulate else r)
        if gbias is not None:
            rb = _v(gy, B, Cout, H, W).sum((0, 2, 3))
            gbias.copy_(gbias + rb if accumulate else rb)
        return 0

    def upconv3x3_wgrad_workspace(self, B, Cin, Cout, H, W):
        return (Cout * Cin * 9 + Cout) * 4

    def upconv3x3_wgrad(self, a, gy, gw, ws, ws_bytes, B, Cin, Cout, H, W, accumulate, gbias=None):
        a_hi = _v(a, B, Cin, H, W).repeat_interleave(2, 2).repeat_interleave(2, 3)
        r = torch.nn.grad.conv2d_weight(a_hi, (Cout, Cin, 3, 3), _v(gy, B, Cout, 2 * H, 2 * W), padding=1)
        gw.copy_(gw + r if accumulate else r)
        if gbias is not None:
            rb = _v(gy, B, Cout, 2 * H, 2 * W).sum((0, 2, 3))
            gbias.copy_(gbias + rb if accumulate else rb)
        return 0

    # theta | phi | g of SelfAttention2d as one pass: the three 1x1 convolutions, their joint input gradient, their filter gradients
    def conv1x1_multi_supported(self, c0, c1, c2, B, Cin, H, W):
        return int((H * W) % 16 == 0 and c0 + c1 + c2 <= 64 and Cin <= 64)

    def conv1x1_multi_fwd(self, x, w, y0, y1, y2, c0, c1, c2, B, Cin, H, W):
        r = F.conv2d(_v(x, B, Cin, H, W), _v(w, c0 + c1 + c2, Cin, 1, 1))
        for y, part in zip((y0, y1, y2), torch.split(r, [c0, c1, c2], 1)):
            y.copy_(part)
        return 0

    def conv1x1_multi_dgrad(self, gy0, gy1, gy2, w, gx, c0, c1, c2, B, Cin, H, W):
        gy = torch.cat([_v(gy0, B, c0, H, W), _v(gy1, B, c1, H, W), _v(gy2, B, c2, H, W)], 1)
        gx.copy_(F.conv_transpose2d(gy, _v(w, c0 + c1 + c2, Cin, 1, 1)))
        return 0

    def conv1x1_multi_wgrad_workspace(self, c0, c1, c2, B, Cin, H, W):
        return 16

    def conv1x1_multi_wgrad(self, x, gy0, gy1, gy2, gw, ws, ws_bytes, c0, c1, c2, B, Cin, H, W, accumulate):
        gy = torch.cat([_v(gy0, B, c0, H, W), _v(gy1, B, c1, H, W), _v(gy2, B, c2, H, W)], 1)
        r = torch.einsum('bchw,bdhw->cd', gy, _v(x, B, Cin, H, W)).reshape(gw.shape)
        gw.copy_(gw + r if accumulate else r)
        return 0

    def conv2d_dgrad(self, gy, w, gx, B, Cin, Cout, H, W, ks):
        gx.copy_(F.conv_transpose2d(_v(gy, B, Cout, H, W), _v(w, Cout, Cin, ks, ks), padding=ks // 2))
        return 0

    def conv2d_wgrad_workspace(self, B, Cin, Cout, H, W, ks):
        return (Cout * Cin * ks * ks + Cout) * 4        # room for one "partial" of the weight and the bias gradient

    def conv2d_wgrad_partials(self, x, gy, ws, ws_bytes, B, Cin, Cout, H, W, ks, want_bias):
        E = Cout * Cin * ks * ks
        assert ws_bytes >= (E + Cout) * 4
        ws[:E].copy_(torch.nn.grad.conv2d_weight(_v(x, B, Cin, H, W), (Cout, Cin, ks, ks), _v(gy, B, Cout, H, W),
                                                 padding=ks // 2).reshape(-1))
        if want_bias:
            ws[E:E + Cout].copy_(_v(gy, B, Cout, H * W).sum((0, 2)))
        return 0

    def conv2d_wgrad_reduce_batch(self, items, n_items):
        import ctypes
        import numpy as np

        def at(addr, n):          # a float32 view of host memory (every tensor is a CPU tensor under the emulator)
            return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_float * n).from_address(addr)))

        assert items.dtype == torch.int64 and tuple(items.shape) == (n_items, 10)
        for part, gw, gbias, B, Cin, Cout, H, W, ks, accumulate in items.tolist():
            E = Cout * Cin * ks * ks
            dst = at(gw, E)
            dst.copy_(dst + at(part, E) if accumulate else at(part, E))
            if gbias:
                dstb = at(gbias, Cout)
                src = at(part + 4 * E, Cout)
                dstb.copy_(dstb + src if accumulate else src)
        return 0

    # stride-2 weight gradients in two steps: stage 1 leaves the finished gradient (and bias gradient) in the workspace
    def poolconv3x3_wgrad_partials(self, x, gy, ws, ws_bytes, B, Cin, Cout, H, W, want_bias):
        E = Cout * Cin * 9
        assert ws_bytes >= (E + Cout) * 4
        gw, gb = torch.zeros(Cout, Cin, 3, 3), torch.zeros(Cout)
        self.poolconv3x3_wgrad(x, gy, gw, None, 0, B, Cin, Cout, H, W, 0, gb if want_bias else None)
        ws[:E].copy_(gw.reshape(-1))
        ws[E:E + Cout].copy_(gb)
        return 0

    def upconv3x3_wgrad_partials(self, a, gy, ws, ws_bytes, B, Cin, Cout, H, W, want_bias):
        E = Cout * Cin * 9
        assert ws_bytes >= (E + Cout) * 4
        gw, gb = torch.zeros(Cout, Cin, 3, 3), torch.zeros(Cout)
        self.upconv3x3_wgrad(a, gy, gw, None, 0, B, Cin, Cout, H, W, 0, gb if want_bias else None)
        ws[:E].copy_(gw.reshape(-1))
        ws[E:E + Cout].copy_(gb)
        return 0

    def s2_wgrad_reduce_batch(self, items, n_items):
        import ctypes
        import numpy as np

        def at(addr, n):
            return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_float * n).from_address(addr)))

        assert items.dtype == torch.int64 and tuple(items.shape) == (n_items, 10)
        for part, gw, gbias, B, Cin, Cout, H, W, mode, accumulate in items.tolist():
            assert mode in (0, 1)
            E = Cout * Cin * 9
            dst = at(gw, E)
            dst.copy_(dst + at(part, E) if accumulate else at(part, E))
            if gbias:
                dstb, src = at(gbias, Cout), at(part + 4 * E, Cout)
                dstb.copy_(dstb + src if accumulate else src)
        return 0

    def rgb_compose_fwd(self, w1, b1, w3, wc, Cout, C, Cimg):
        w1c = torch.cat([_v(w1, C, Cimg), _v(b1, C, 1)], 1)
        _v(wc, Cout, Cimg + 1, 9).copy_(torch.einsum('mc,omt->oct', w1c, _v(w3, Cout, C, 9)))
        return 0

    def rgb_compose_bwd(self, gwc, w1, b1, w3, gw1, gb1, gw3, Cout, C, Cimg, accumulate):
        w1c = torch.cat([_v(w1, C, Cimg), _v(b1, C, 1)], 1)
        g = _v(gwc, Cout, Cimg + 1, 9)
        r3 = torch.einsum('oct,mc->omt', g, w1c).reshape(gw3.shape)
        r1c = torch.einsum('oct,omt->mc', g, _v(w3, Cout, C, 9))
        r1, rb = r1c[:, :Cimg].reshape(gw1.shape), r1c[:, Cimg].reshape(gb1.shape)
        gw3.copy_(gw3 + r3 if accumulate else r3)
        gw1.copy_(gw1 + r1 if accumulate else r1)
        gb1.copy_(gb1 + rb if accumulate else rb)
        return 0

    def poolconv3x3_weights_batch(self, items, n_items):
        import ctypes
        import numpy as np

        def at(addr, n):
            return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_float * n).from_address(addr)))

        assert items.dtype == torch.int64 and tuple(items.shape) == (n_items, 5)
        for w, w4, wp, Cout, Cin in items.tolist():
            self.poolconv3x3_weights(at(w, Cout * Cin * 9), at(w4, Cout * Cin * 16), at(wp, Cout * Cin * 16), Cout, Cin)
        return 0

    def conv2d_wgrad(self, x, gy, gw, gbias, ws, ws_bytes, B, Cin, Cout, H, W, ks, accumulate):
        r = torch.nn.grad.conv2d_weight(_v(x, B, Cin, H, W), (Cout, Cin, ks, ks), _v(gy, B, Cout, H, W), padding=ks // 2)
        gw.copy_(gw + r if accumulate else r)
        if gbias is not None:
            rb = _v(gy, B, Cout, H * W).sum((0, 2))
            gbias.copy_(gbias + rb if accumulate else rb)
        return 0

    def channel_sum(self, x, out, ws, B, C, HW, accumulate):
        r = _v(x, B, C, HW).sum((0, 2))
        out.copy_(out + r if accumulate else r)
        return 0

    def channel_bcast(self, v, out, B, C, HW):
        _v(out, B, C, HW).copy_(_v(v, 1, C, 1).expand(B, C, HW))
        return 0

    # ---------------------------------------------------------------- gemm
    def gemm(self, A, Bm, C, bias, M, N, K, lda, ldb, ldc, ta, tb, batch, sA, sB, sC, beta):
        a = A.reshape(batch, -1, lda)
        b = Bm.reshape(batch, -1, ldb)
        a = a.transpose(1, 2) if ta else a
        b = b.transpose(1, 2) if tb else b
        r = torch.bmm(a, b)
        if bias is not None:
            r = r + bias.view(1, 1, N)
        if beta != 0:
            r = r + beta * C.view(batch, M, N)
        C.view(batch, M, N).copy_(r)
        return 0

    # ---------------------------------------------------------------- batch norm
    def bn_workspace(self, B, C, HW):
        return 16

    def bn_train_stats(self, x, mean, invstd, rm, rv, nbt, momentum, eps, ws, B, C, HW, replicate=1):
        if nbt is not None:
            nbt.add_(1)
        xv = _v(x, B, C, HW)
        n = B * HW * replicate
        m = xv.mean((0, 2))
        var = xv.var((0, 2), unbiased=False)
        mean.copy_(m)
        invstd.copy_(1 / torch.sqrt(var + eps))
        if rm is not None:
            rm.mul_(1 - momentum).add_(momentum * m)
            rv.mul_(1 - momentum).add_(momentum * var * (n / max(n - 1, 1)))
        return 0

    def bn_train_fwd(self, x, mean, invstd, rm, rv, nbt, gamma, beta, slope, momentum, eps, z, ws, B, C, HW, replicate=1):
        self.bn_train_stats(x, mean, invstd, rm, rv, nbt, momentum, eps, ws, B, C, HW, replicate)
        return self.bn_act_fwd(x, mean, invstd, gamma, beta, slope, z, B, C, HW)

    def bn_eval_stats(self, rm, rv, mean, invstd, eps, C):
        mean.copy_(rm)
        invstd.copy_(1 / torch.sqrt(rv + eps))
        return 0

    @staticmethod
    def _bn_parts(x, mean, invstd, gamma, beta, slope, B, C, HW):
        xv = _v(x, B, C, HW)
        xhat = (xv - mean.view(1, C, 1)) * invstd.view(1, C, 1)
        y = xhat * gamma.view(1, C, 1) + beta.view(1, C, 1)
        s = torch.where(y >= 0, torch.ones_like(y), torch.full_like(y, slope))
        return xhat, y, s

    def bn_act_fwd(self, x, mean, invstd, gamma, beta, slope, z, B, C, HW):
        xhat, y, s = self._bn_parts(x, mean, invstd, gamma, beta, slope, B, C, HW)
        _v(z, B, C, HW).copy_(y * s)
        return 0

    def bn_act_bwd(self, gz, x, mean, invstd, gamma, beta, slope, training, gx, gg, gb, ws, B, C, HW, accumulate, gx_add=None):
        xhat, y, s = self._bn_parts(x, mean, invstd, gamma, beta, slope, B, C, HW)
        gyh = _v(gz, B, C, HW) * s
        n = B * HW
        sb = gyh.sum((0, 2))
        sg = (gyh * xhat).sum((0, 2))
        gg.copy_(gg + sg if accumulate else sg)
        gb.copy_(gb + sb if accumulate else sb)
        if gx is not None:
            k = (gamma * invstd).view(1, C, 1)
            if training:
                r = k * (gyh - sb.view(1, C, 1) / n - xhat * sg.view(1, C, 1) / n)
            else:
                r = k * gyh
            if gx_add is not None:
                r = r + _v(gx_add, B, C, HW)
            _v(gx, B, C, HW).copy_(r)
        return 0

    # grouped forms = the single-group calls on each group's images, one after the other (what two separate forwards /
    # backwards of the reference do: trainers/cnn.py:122-123)
    def bn_train_fwd_groups(self, x, mean, invstd, rm, rv, nbt, gamma, beta, slope, momentum, eps, z, ws, G, B, C, HW, replicate=1):
        xv, zv = _v(x, G, B * C * HW), _v(z, G, B * C * HW)
        for g in range(G):
            self.bn_train_fwd(xv[g], mean[g * C:(g + 1) * C], invstd[g * C:(g + 1) * C], rm, rv, nbt, gamma, beta, slope, momentum,
                              eps, zv[g], ws, B, C, HW, replicate)
        return 0

    def bn_act_bwd_groups(self, gz, x, mean, invstd, gamma, beta, slope, training, gx, gg, gb, ws, G, B, C, HW, accumulate,
                          gx_add=None, add_groups=1):
        gzv, xv = _v(gz, G, B * C * HW), _v(x, G, B * C * HW)
        gxv = None if gx is None else _v(gx, G, B * C * HW)
        addv = None if gx_add is None else _v(gx_add, add_groups, B * C * HW)
        for g in range(G):
            self.bn_act_bwd(gzv[g], xv[g], mean[g * C:(g + 1) * C], invstd[g * C:(g + 1) * C], gamma, beta, slope, training,
                            None if gxv is None else gxv[g], gg, gb, ws, B, C, HW, accumulate if g == 0 else 1,
                            addv[g] if (addv is not None and g < add_groups) else None)
        return 0

    def bn_act_dbwd(self, v, vg, vb, gz, x, mean, invstd, gamma, beta, slope, a_gz, a_x, a_gamma, ws, B, C, HW, accumulate=0):
        xhat, y, s = self._bn_parts(x, mean, invstd, gamma, beta, slope, B, C, HW)
        n = B * HW
        c = lambda t: t.view(1, C, 1)
        gyh = _v(gz, B, C, HW) * s
        vv = _v(v, B, C, HW)
        vg = torch.zeros_like(gamma) if vg is None else vg
        vb = torch.zeros_like(gamma) if vb is None else vb
        S1, S2 = vv.sum((0, 2)), (vv * xhat).sum((0, 2))
        S3, S4 = gyh.sum((0, 2)), (gyh * xhat).sum((0, 2))
        S5 = (vv * gyh).sum((0, 2))
        r = invstd
        # adjoint of gyh: gamma r P(v) + vgamma xhat + vbeta
        Pv = vv - c(S1) / n - xhat * c(S2) / n
        a_gyh = c(gamma * r) * Pv + c(vg) * xhat + c(vb)
        _v(a_gz, B, C, HW).copy_(a_gyh * s)
        # adjoint of gamma: r * sum v P(gyh)
        A = S5 - S1 * S3 / n - S2 * S4 / n
        a_gamma.copy_(a_gamma + r * A if accumulate else r * A)
        # adjoint of x
        cg, cv = S4 / n, S2 / n
        q = -c(gamma * r) * (c(cg) * vv + c(cv) * gyh) + c(vg) * gyh
        qm = q.sum((0, 2)) / n
        qx = (q * xhat).sum((0, 2)) / n
        Pq = q - c(qm) - xhat * c(qx)
        _v(a_x, B, C, HW).copy_(c(r) * Pq - c(gamma * A * r * r / n) * xhat)
        return 0

    # ---- SyncBN protocol (local sums -> all-reduce by the caller -> finish), float64 sums [C][K]
    def bn_sync_stats_local(self, x, sums, ws, B, C, HW):
        xv = _v(x, B, C, HW).double()
        m = xv.mean((0, 2))
        var = xv.var((0, 2), unbiased=False)
        sums.view(C, 3).copy_(torch.stack([m, m * m, var], 1))
        return 0

    def bn_sync_stats_finish(self, sums, world, mean, invstd, rm, rv, nbt, momentum, eps, count_global, replicate, C):
        if nbt is not None:
            nbt.add_(1)
        s = sums.view(C, 3) / world
        m = s[:, 0]
        var = (s[:, 2] + (s[:, 1] - m * m)).clamp_min(0)
        mean.copy_(m.float())
        invstd.copy_((1 / torch.sqrt(var + eps)).float())
        if rm is not None:
            n = float(count_global * replicate)
            rm.copy_(((1 - momentum) * rm.double() + momentum * m).float())
            rv.copy_(((1 - momentum) * rv.double() + momentum * var * (n / max(n - 1, 1))).float())
        return 0

    def bn_sync_bwd_local(self, gz, x, mean, invstd, gamma, beta, slope, sums, ws, B, C, HW):
        xhat, y, s = self._bn_parts(x, mean, invstd, gamma, beta, slope, B, C, HW)
        gyh = (_v(gz, B, C, HW) * s).double()
        sums.view(C, 2).copy_(torch.stack([gyh.sum((0, 2)), (gyh * xhat.double()).sum((0, 2))], 1))
        return 0

    def bn_sync_bwd_finish(self, gz, x, mean, invstd, gamma, beta, slope, local, glob, count_global, gx, gg, gb, ws, B, C, HW,
                           accumulate, gx_add=None):
        xhat, y, s = self._bn_parts(x, mean, invstd, gamma, beta, slope, B, C, HW)
        gyh = _v(gz, B, C, HW) * s
        lo, gl = local.view(C, 2), glob.view(C, 2)
        sb, sg = lo[:, 0].float(), lo[:, 1].float()
        gg.copy_(gg + sg if accumulate else sg)
        gb.copy_(gb + sb if accumulate else sb)
        if gx is not None:
            k1 = (gl[:, 0] / count_global).float().view(1, C, 1)
            k2 = (gl[:, 1] / count_global).float().view(1, C, 1)
            r = (gamma * invstd).view(1, C, 1) * (gyh - k1 - xhat * k2)
            _v(gx, B, C, HW).copy_(r if gx_add is None else r + _v(gx_add, B, C, HW))
        return 0

    def bn_sync_dbwd_local(self, v, gz, x, mean, invstd, gamma, beta, slope, sums, ws, B, C, HW):
        xhat, y, s = self._bn_parts(x, mean, invstd, gamma, beta, slope, B, C, HW)
        xhat = xhat.double()
        gyh = (_v(gz, B, C, HW) * s).double()
        vv = _v(v, B, C, HW).double()
        cols = [vv.sum((0, 2)), (vv * xhat).sum((0, 2)), gyh.sum((0, 2)), (gyh * xhat).sum((0, 2)), (vv * gyh).sum((0, 2))]
        sums.view(C, 5).copy_(torch.stack(cols, 1))
        return 0

    def bn_sync_dbwd_finish(self, v, gz, x, mean, invstd, gamma, beta, slope, glob, count_global, world, a_gz, a_x, a_gamma,
                            ws, B, C, HW, accumulate=0):
        xhat, y, s = self._bn_parts(x, mean, invstd, gamma, beta, slope, B, C, HW)
        n = float(count_global)
        c = lambda t: t.float().view(1, C, 1)
        gyh = _v(gz, B, C, HW) * s
        vv = _v(v, B, C, HW)
        S1, S2, S3, S4, S5 = glob.view(C, 5).unbind(1)
        r, g = invstd.double(), gamma.double()
        A = S5 - S1 * S3 / n - S2 * S4 / n
        cg, cv = S4 / n, S2 / n
        qm = -g * r * (cg * S1 / n + cv * S3 / n)
        qx = -g * r * (cg * S2 / n + cv * S4 / n)
        a_gamma.copy_(a_gamma + (r * A / world).float() if accumulate else (r * A / world).float())
        Pv = vv - c(S1 / n) - xhat * c(S2 / n)
        _v(a_gz, B, C, HW).copy_(c(g * r) * Pv * s)
        q = -c(g * r) * (c(cg) * vv + c(cv) * gyh)
        _v(a_x, B, C, HW).copy_(c(r) * (q - c(qm) - xhat * c(qx)) - c(g * A * r * r / n) * xhat)
        return 0

    # ---------------------------------------------------------------- input pipeline
    def image_bytes_batch(self, archive, index, oy, ox, out, B, n_images, H, W, channels, size):
        imgs = archive.view(n_images, H, W, channels)
        for b in range(B):
            y0 = int(oy[b]) if oy is not None else 0
            x0 = int(ox[b]) if ox is not None else 0
            crop = imgs[int(index[b]), y0:y0 + size, x0:x0 + size]
            t = crop.permute(2, 0, 1).contiguous().to(torch.float32).div(255)       # ToTensor
            out.view(B, channels, size, size)[b].copy_(t.sub_(0.5).div_(0.5))          # Normalize(.5, .5)
        return 0

    # ---------------------------------------------------------------- FID / IS math
    def gemm_big_supported(self, M, N, K, lda, ldb, transA):
        return int(lda % 4 == 0 and ldb % 4 == 0 and (transA or K % 4 == 0))

    def gemm_big(self, A, B, C, M, N, K, lda, ldb, ldc, transA, alpha, diag):
        a = A.view(-1)[:(K if transA else M) * lda].view(-1, lda)
        a = a[:, :M].t() if transA else a[:, :K]
        r = alpha * (a @ B.view(-1, ldb)[:K, :N])
        if diag:
            r = r + diag * torch.eye(M, N)
        C.view(-1, ldc)[:M, :N].copy_(r)
        return 0

    def center_rows(self, X, mean, N, D):
        X.view(N, D).sub_(mean.view(1, D))
        return 0

    def trace(self, A, out, D, ld):
        out.copy_(torch.diagonal(A.view(-1, ld)[:D, :D]).double().sum().float())
        return 0

    def is_kl_rows(self, p, mean, rows, N, Cn):
        pv = p.view(N, Cn)
        rows.copy_((pv * (pv.log() - mean.view(1, Cn).log())).sum(1))
        return 0

    # ---------------------------------------------------------------- resampling
    def up2x(self, x, y, alpha, BC, H, W):
        _v(y, BC, 2 * H, 2 * W).copy_(alpha * _v(x, BC, H, W).repeat_interleave(2, 1).repeat_interleave(2, 2))
        return 0

    def pool2(self, x, residual, y, alpha, BC, H, W):
        r = alpha * 4 * F.avg_pool2d(_v(x, 1, BC, H, W), 2)[0]
        _v(y, BC, H // 2, W // 2).copy_(r if residual is None else _v(residual, BC, H // 2, W // 2) + r)
        return 0

    def bilinear_half_fwd(self, x, y, BC, H, W):
        _v(y, BC, H // 2, W // 2).copy_(F.interpolate(_v(x, 1, BC, H, W), scale_factor=0.5, mode='bilinear',
                                                      align_corners=True)[0])
        return 0

    def bilinear_half_bwd(self, gy, residual, gx, BC, H, W):
        with torch.enable_grad():
            xin = torch.zeros(1, BC, H, W, requires_grad=True)
            out = F.interpolate(xin, scale_factor=0.5, mode='bilinear', align_corners=True)
            g, = torch.autograd.grad(out, xin, _v(gy, 1, BC, H // 2, W // 2).detach())
        _v(gx, BC, H, W).copy_(g[0] if residual is None else _v(residual, BC, H, W) + g[0])
        return 0

    def inception_preprocess(self, x, mean, stdv, out, B, C, H, W, OH, OW, stages):
        v = _v(x, B, C, H, W)
        for _ in range(stages):
            v = (v + 1.) / 2.0
            v = (v - mean.view(1, C, 1, 1)) / stdv.view(1, C, 1, 1)
        if (H, W) != (OH, OW):
            v = F.interpolate(v, size=(OH, OW), mode='bilinear', align_corners=True)
        _v(out, B, C, OH, OW).copy_(v)
        return 0

    def maxpool2_fwd(self, x, y, idx, BC, H, W):
        xv = _v(x, BC, H // 2, 2, W // 2, 2).permute(0, 1, 3, 2, 4).reshape(BC, H // 2, W // 2, 4)
        # first maximum in window order (NaN-free inputs)
        m = xv.max(-1, keepdim=True).values
        first = (xv == m).to(torch.uint8).argmax(-1)
        _v(y, BC, H // 2, W // 2).copy_(m.squeeze(-1))
        _v(idx, BC, H // 2, W // 2).copy_(first.to(torch.uint8))
        return 0

    def maxpool2_bwd(self, gy, idx, gx, BC, H, W):
        oh, ow = H // 2, W // 2
        onehot = F.one_hot(_v(idx, BC, oh, ow).long(), 4).to(gy.dtype) * _v(gy, BC, oh, ow, 1)
        _v(gx, BC, H, W).copy_(onehot.view(BC, oh, ow, 2, 2).permute(0, 1, 3, 2, 4).reshape(BC, H, W))
        return 0

    def maxpool2_gather(self, x, idx, y, BC, H, W):
        oh, ow = H // 2, W // 2
        xv = _v(x, BC, oh, 2, ow, 2).permute(0, 1, 3, 2, 4).reshape(BC, oh, ow, 4)
        _v(y, BC, oh, ow).copy_(xv.gather(-1, _v(idx, BC, oh, ow, 1).long()).squeeze(-1))
        return 0

    # ---------------------------------------------------------------- row ops
    def row_sum(self, x, out, alpha, rows, cols):
        out.view(rows).copy_(alpha * _v(x, rows, cols).sum(1))
        return 0

    def row_bcast(self, v, out, alpha, rows, cols):
        _v(out, rows, cols).copy_(alpha * v.reshape(rows, 1).expand(rows, cols))
        return 0

    def repeat_rows(self, x, out, alpha, rows, cols, reps):
        _v(out, reps, rows, cols).copy_(alpha * _v(x, 1, rows, cols).expand(reps, rows, cols))
        return 0

    def sum_reps(self, x, out, alpha, rows, cols, reps):
        _v(out, rows, cols).copy_(alpha * _v(x, reps, rows, cols).sum(0))
        return 0

    def repeat_rows_groups(self, x, out, alpha, rows, cols, reps, groups):
        _v(out, groups, reps, rows, cols).copy_(alpha * _v(x, groups, 1, rows, cols).expand(groups, reps, rows, cols))
        return 0

    def sum_reps_groups(self, x, out, alpha, rows, cols, reps, groups):
        _v(out, groups, rows, cols).copy_(alpha * _v(x, groups, reps, rows, cols).sum(1))
        return 0

    # ---------------------------------------------------------------- elementwise
    def add(self, a, b, out, n):
        out.copy_(a + b)
        return 0

    def add4(self, a, b, c, d, out, n):
        r = a + b
        if c is not None:
            r = r + c
        if d is not None:
            r = r + d
        out.copy_(r)
        return 0

    def mul(self, a, b, out, n):
        out.copy_(a * b)
        return 0

    def scale(self, x, alpha, out, n):
        out.copy_(x * alpha)
        return 0

    def scale_dev(self, s, alpha, x, out, n):
        out.copy_((alpha * s.reshape(())) * x)
        return 0

    def scale_add_dev(self, s, a, b, out, n):
        out.copy_(s.reshape(()) * a + b)
        return 0

    def reduce_workspace(self, n):
        return 16

    def dot(self, a, b, alpha, out, ws, n, accumulate):
        r = alpha * (a.double() * b.double()).sum().to(a.dtype)
        out.copy_(out + r if accumulate else r)
        return 0

    def lrelu_bwd(self, g, x, slope, out, n):
        out.copy_(torch.where(x >= 0, g, g * slope))
        return 0

    def elu_fwd(self, x, alpha, scale, y, n):
        y.copy_(scale * F.elu(x, alpha))
        return 0

    def elu_bwd(self, g, x, alpha, scale, order, out, n):
        neg = g * (alpha * scale) * torch.exp(x)
        pos = g * scale if order == 1 else torch.zeros_like(g)
        out.copy_(torch.where(x <= 0, neg, pos))
        return 0

    def tanh_fwd(self, x, y, n):
        y.copy_(torch.tanh(x))
        return 0

    def tanh_bwd(self, g, y, out, n):
        out.copy_(g * (1 - y * y))
        return 0

    def copy_channels(self, src, dst, B, Cs, Cd, HW, fill):
        sv, dv = _v(src, B, Cs, HW), _v(dst, B, Cd, HW)
        k = min(Cs, Cd)
        dv[:, :k].copy_(sv[:, :k])
        if Cd > Cs:
            dv[:, Cs:].fill_(fill)
        return 0

    def fill(self, x, value, n):
        x.fill_(value)
        return 0

    # ---------------------------------------------------------------- softmax
    def softmax_fwd(self, s, y, rows, cols):
        _v(y, rows, cols).copy_(F.softmax(_v(s, rows, cols), -1))
        return 0

    def softmax_bwd(self, gy, y, gs, rows, cols):
        g, yy = _v(gy, rows, cols), _v(y, rows, cols)
        _v(gs, rows, cols).copy_(yy * (g - (g * yy).sum(-1, keepdim=True)))
        return 0

    def softmax_dbwd(self, v, gy, y, out, rows, cols):
        vv, g, yy = _v(v, rows, cols), _v(gy, rows, cols), _v(y, rows, cols)
        d = (g * yy).sum(-1, keepdim=True)
        e = (vv * yy).sum(-1, keepdim=True)
        _v(out, rows, cols).copy_(vv * g - vv * d - g * e)
        return 0

    # ---------------------------------------------------------------- fused attention core
    def attn_supported(self, D, DV):
        return int((D, DV) in ((1, 4), (2, 8), (4, 16), (8, 32), (16, 64)))

    def attn_bwd_workspace(self, B, D, DV, N, M):
        return 16

    def attn_fwd(self, theta, phi, g, o, lse, B, D, DV, N, M):
        s = torch.bmm(theta.view(B, D, N).transpose(1, 2), phi.view(B, D, M))
        lse.view(B, N).copy_(torch.logsumexp(s, -1))
        o.view(B, DV, N).copy_(torch.bmm(g.view(B, DV, M), F.softmax(s, -1).transpose(1, 2)))
        return 0

    def attn_bwd(self, go, theta, phi, g, o, lse, dtheta, dphi, dg, ws, B, D, DV, N, M):
        with torch.enable_grad():
            t, p, gg = (a.detach().clone().requires_grad_() for a in (theta.view(B, D, N), phi.view(B, D, M), g.view(B, DV, M)))
            out = torch.bmm(gg, F.softmax(torch.bmm(t.transpose(1, 2), p), -1).transpose(1, 2))
            a, b, c = torch.autograd.grad(out, (t, p, gg), go.view(B, DV, N).detach())
        dtheta.view(B, D, N).copy_(a); dphi.view(B, D, M).copy_(b); dg.view(B, DV, M).copy_(c)
        return 0

    def attn_dbwd_rows(self, s, lse, gp, u, v, rows, cols):
        S, G, U, V = _v(s, rows, cols), _v(gp, rows, cols), _v(u, rows, cols), _v(v, rows, cols)
        P = torch.exp(S - lse.reshape(rows, 1))
        d = (P * G).sum(-1, keepdim=True)
        e = (P * U).sum(-1, keepdim=True)
        dP = V + U * (G - d) - G * e
        z = (P * dP).sum(-1, keepdim=True)
        gS, dG, dS = P * (G - d), P * (U - e), P * (dP - z)
        S.copy_(P); G.copy_(gS); U.copy_(dG); V.copy_(dS)
        return 0

    def attn_dbwd_supported(self, D, DV, M):
        cpl = (M + 63) // 64
        return int(bool(self.attn_supported(D, DV)) and M > 0 and cpl <= 4 and 3 * (D + DV) * cpl <= 256)

    def attn_dbwd_workspace(self, B, D, DV, N, M):
        return 16

    def attn_dbwd(self, go, theta, phi, g, lse, a, b, c, d_go, d_theta, d_phi, d_g, ws, B, D, DV, N, M):
        assert self.attn_dbwd_supported(D, DV, M)
        with torch.enable_grad():
            gO, t, p, gg = (x.detach().clone().requires_grad_() for x in
                            (go.view(B, DV, N), theta.view(B, D, N), phi.view(B, D, M), g.view(B, DV, M)))
            out = torch.bmm(gg, F.softmax(torch.bmm(t.transpose(1, 2), p), -1).transpose(1, 2))
            first = torch.autograd.grad(out, (t, p, gg), gO, create_graph=True)
            second = torch.autograd.grad(first, (gO, t, p, gg), (a.view(B, D, N), b.view(B, D, M), c.view(B, DV, M)))
        for dst, src in zip((d_go, d_theta, d_phi, d_g), second):
            dst.view(src.shape).copy_(src)
        return 0

    # ---------------------------------------------------------------- iqn / losses
    def iqn_cos_embed(self, taus, rng, out, n, dims):
        out.copy_(torch.cos(taus.view(n, 1).repeat(1, dims) * math.pi * rng))
        return 0

    def iqn_loss(self, preds, target, taus, k, loss, dpreds, ws, Q, B):
        p = preds.view(Q, B)
        err = target.view(1, B) - p
        a = err.abs()
        hub = torch.where(a <= k, 0.5 * err * err, k * (a - 0.5 * k))
        dhub = torch.where(a <= k, err, k * torch.sign(err))        # d huber / d err
        wgt = (taus.view(Q, B) - (err < 0).float()).abs()
        loss.copy_((wgt * hub).sum(0).mean())
        dpreds.view(Q, B).copy_(-wgt * dhub / B)
        return 0

    def iqn_loss_groups(self, preds, target, taus, k, loss, dpreds, ws, Q, B, groups):
        total = torch.zeros(())
        one = torch.zeros(())
        for g in range(groups):
            rows = slice(g * Q * B, (g + 1) * Q * B)
            self.iqn_loss(preds.reshape(-1)[rows], target.reshape(-1)[g * B:(g + 1) * B], taus.reshape(-1)[rows], k, one,
                          dpreds.view(-1)[rows], ws, Q, B)
            total = total + one
        loss.copy_(total)
        return 0

    def bce_logits(self, logits, targets, loss, dlogits, ws, n):
        x, t = logits.view(-1), targets.view(-1)
        loss.copy_(F.binary_cross_entropy_with_logits(x, t))
        dlogits.view(-1).copy_((torch.sigmoid(x) - t) / n)
        return 0

    def sumsq(self, x, alpha, out, ws, n):
        out.copy_(alpha * x.pow(2).sum())
        return 0

    # ---------------------------------------------------------------- spectral norm
    def sn_power_iter(self, W, u, v, sigma, rows, cols, n_iter, eps):
        Wm = W.view(rows, cols)
        for _ in range(n_iter):
            F.normalize(torch.mv(Wm.t(), u), dim=0, eps=eps, out=v)
            F.normalize(torch.mv(Wm, v), dim=0, eps=eps, out=u)
        if sigma is not None:
            sigma.copy_(torch.dot(u, torch.mv(Wm, v)))
        return 0

    def recip(self, x, out, n):
        out.copy_(1.0 / x)
        return 0

    # ---------------------------------------------------------------- optimiser
    def adam_step(self, p, g, m, v, hyper, eps, n):
        step_size, bc2_sqrt, b1, b2, omb1, omb2 = [float(h) for h in hyper]
        m.lerp_(g, omb1)
        v.mul_(b2).addcmul_(g, g, value=omb2)
        denom = (v.sqrt() / bc2_sqrt).add_(eps)
        p.addcdiv_(m, denom, value=-step_size)
        return 0

    def ema(self, t, p, lr, n):
        t.add_((p - t) * lr)
        return 0
