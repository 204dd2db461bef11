"""Numpy tile backend for style_transfer2_amd.tiled.TiledTransfer, built on the CPU oracle.  TEST INFRASTRUCTURE.

It restates, for ONE tile of a sharded image, exactly what oracle.TransferOracle does for the whole image
(worker.py:231-310, utils.py:285-304, optimizers.py:20-27), with every reduction restricted to the tile's region
of each blob and normalised by the GLOBAL element counts.  Used to validate the distributed algorithm on the CPU
(gloo) against the single-process oracle, and as the checker of the HIP tile backend.
"""
import numpy as np
import torch

import oracle
from oracle.objective import weight_table
from style_transfer2_amd.tiling import blob_geometry

F32 = np.float32
EPS_W = 1e-15


class OracleTileBackend:
    def __init__(self, topology, net_params, grid, rank, content, style, init, weights, params, step_size=10):
        self.net = oracle.NetOracle(topology, net_params)
        self.grid, self.rank = grid, rank
        self.win, self.tile = grid.windows[rank], grid.tiles[rank]
        w, t = self.win, self.tile
        crop = lambda im, r: np.ascontiguousarray(im[r.y0:r.y1, r.x0:r.x1])
        x = self.net.preprocess(crop(init, w))
        self.x = [x.copy(), x.copy()]
        self.cur = 0
        self.names = self.net.layers()
        self.local_geo = blob_geometry(topology, w.y1 - w.y0, w.x1 - w.x0)
        self.global_geo = blob_geometry(topology, grid.gH, grid.gW)
        self.content_feats = {k: v.copy() for k, v in self.net.forward(self.net.preprocess(crop(content, w))).items()}
        self.grams = {k: oracle.gram(v) for k, v in self.net.forward(self.net.preprocess(style)).items()}
        rows, cells = weight_table(weights)
        self.active = []
        for n in rows:
            cw, sw, dw = (cells[k][n] for k in ('content', 'style', 'deepdream'))
            flags = tuple(abs(v) > EPS_W for v in (cw, sw, dw))
            if any(flags):
                self.active.append((n, cw, sw, dw) + flags)
        self.params = params
        self.step_size = step_size
        self.norms = {}
        th, tw = t.y1 - t.y0, t.x1 - t.x0
        self.m = np.zeros((3, th, tw), F32)
        self.v = np.zeros((3, th, tw), F32)
        self.items1 = self.items2 = 0

    # ---- helpers ---------------------------------------------------------------------------------------
    def _roi(self, name):
        i = self.names.index(name)
        c, h, w, s = self.local_geo[i]
        r = self.grid.roi_in_blob(self.rank, (h, w), s)
        return (slice(None), slice(r.y0, r.y1), slice(r.x0, r.x1))

    def _n_global(self, name):
        c, h, w, _ = self.global_geo[self.names.index(name)]
        return c * h * w

    def x_cur(self):
        return torch.from_numpy(self.x[self.cur][0])

    def x_next(self):
        return torch.from_numpy(self.x[self.cur ^ 1][0])

    def swap(self):
        self.cur ^= 1

    # ---- phase 1 ---------------------------------------------------------------------------------------
    def forward_partials(self):
        names = [a[0] for a in self.active]
        self.feats = self.net.forward(self.x[self.cur], names)
        parts = []
        for name, cw, sw, dw, c_on, s_on, d_on in self.active:
            F = self.feats[name][0]
            roi = self._roi(name)
            Fr = F[roi]
            n = self._n_global(name)
            sums = np.zeros(4, np.float64)
            if c_on:
                d = Fr - self.content_feats[name][0][roi]
                gc = F32(2 / n) * d
                sums[0], sums[1] = np.sum(d.astype(np.float64)**2), np.sum(gc.astype(np.float64)**2)
            if d_on:
                gd = F32(-2 / n) * Fr
                sums[2], sums[3] = np.sum(Fr.astype(np.float64)**2), np.sum(gd.astype(np.float64)**2)
            parts.append(sums.astype(F32))
            if s_on:
                f2 = Fr.reshape(Fr.shape[0], -1)
                parts.append(np.dot(f2, f2.T).ravel())
        self._p1 = np.concatenate(parts) if parts else np.zeros(0, F32)
        return torch.from_numpy(self._p1)

    def _unpack_p1(self):
        out, pos = {}, 0
        for name, cw, sw, dw, c_on, s_on, d_on in self.active:
            sums = self._p1[pos:pos + 4]
            pos += 4
            graw = None
            if s_on:
                c = self.feats[name].shape[1]
                graw = self._p1[pos:pos + c * c].reshape(c, c)
                pos += c * c
            out[name] = (sums, graw)
        return out

    # ---- phase 2 ---------------------------------------------------------------------------------------
    def losses_need_style_norm(self):
        self.red = self._unpack_p1()
        self.sgrad, self.D, s2 = {}, {}, []
        missing = False
        for name, cw, sw, dw, c_on, s_on, d_on in self.active:
            n = self._n_global(name)
            sums, graw = self.red[name]
            if c_on and ('c', name) not in self.norms:
                self.norms['c', name] = np.sqrt(F32(sums[1] / n))
            if d_on and ('d', name) not in self.norms:
                self.norms['d', name] = np.sqrt(F32(sums[3] / n))
            if s_on:
                roi = self._roi(name)
                Fr = self.feats[name][0][roi]
                c = Fr.shape[0]
                D = graw / F32(n) - self.grams[name]
                S = np.dot(D, Fr.reshape(c, -1)).reshape(Fr.shape)
                S *= 2 / (D.size * n)
                self.D[name], self.sgrad[name] = D, S
                s2.append(np.sum(S.astype(np.float64)**2))
                missing = missing or ('s', name) not in self.norms
        self._s2 = np.asarray(s2, F32)
        self._s2_reduced = missing
        return torch.from_numpy(self._s2) if missing else None

    def finish_losses(self):
        k = 0
        self.diffs = {}
        for name, cw, sw, dw, c_on, s_on, d_on in self.active:
            n = self._n_global(name)
            F = self.feats[name]
            roi = (slice(None),) + self._roi(name)
            acc = np.zeros_like(F)
            if c_on:
                d = F[roi] - self.content_feats[name][roi]
                acc[roi] += cw * (F32(2 / n) * d) / self.norms['c', name]
            if s_on:
                if ('s', name) not in self.norms:
                    self.norms['s', name] = np.sqrt(F32(self._s2[k] / n))
                acc[roi] += (sw / self.norms['s', name]) * self.sgrad[name][None]
                k += 1
            if d_on:
                acc[roi] += dw * (F32(-2 / n) * F[roi]) / self.norms['d', name]
            self.diffs[name] = acc

    # ---- phase 3 ---------------------------------------------------------------------------------------
    def backward(self):
        self.g = self.net.backward(self.diffs)[0].copy()
        return torch.from_numpy(self.g)

    # ---- phase 4 ---------------------------------------------------------------------------------------
    def _tile_slices(self):
        t, w = self.tile, self.win
        return slice(t.y0 - w.y0, t.y1 - w.y0), slice(t.x0 - w.x0, t.x1 - w.x0)

    # ---- L-BFGS pieces (tiled.TiledTransfer, optimizer='lbfgs'): plain fp32 numpy on the tile ------------------------------
    def gradient(self, ring):
        self._no_update = True
        try:
            return self.update(ring)
        finally:
            self._no_update = False

    def grad_tile(self):
        return torch.from_numpy(self._grad_tile.copy())

    def vcopy(self, v):
        return v.clone()

    def vdot(self, a, b):
        from scipy.linalg import blas
        return torch.from_numpy(np.asarray([blas.sdot(a.numpy().ravel(), b.numpy().ravel())], F32))      # utils.dot on this rank's part

    def vaxpy(self, alpha, x, y):
        y.numpy()[...] = F32(alpha) * x.numpy() + y.numpy()

    def vscale(self, alpha, y):
        y.numpy()[...] = F32(alpha) * y.numpy()

    def vdiv(self, divisor, y):
        y.numpy()[...] = (y.numpy().astype(np.float64) / float(divisor)).astype(F32)      # p /= np.float64 scalar

    def apply_step(self, s):
        ys, xs = self._tile_slices()
        nxt = self.x[self.cur ^ 1]
        nxt[...] = self.x[self.cur]
        nxt[0][:, ys, xs] = self.x[self.cur][0][:, ys, xs] + s.numpy()

    def update(self, ring):
        t, w = self.tile, self.win
        ys, xs = slice(t.y0 - w.y0, t.y1 - w.y0), slice(t.x0 - w.x0, t.x1 - w.x0)
        xt = self.x[self.cur][0][:, ys, xs]
        U = ring.numpy().copy()
        U[:, 1:-1, 1:-1] = xt
        U = U / 255
        beta, ppow = self.params['tv_power'], self.params['p_power']
        a = U[:, :-1, :-1] - U[:, :-1, 1:]          # a[y,x] = u[y,x] - u[y,x+1], y,x in 0..th / 0..tw
        b = U[:, :-1, :-1] - U[:, 1:, :-1]
        q = a**2 + b**2 + 1e-8
        kk = (beta / 2) * q**(beta / 2 - 1)
        da, db = 2 * a * kk, 2 * b * kk
        g_tv = da[:, 1:, 1:] + db[:, 1:, 1:]
        g_tv = g_tv - da[:, 1:, :-1]
        g_tv = g_tv - db[:, :-1, 1:]
        u = U[:, 1:-1, 1:-1]
        tv_val = np.sum((q[:, 1:, 1:]**(beta / 2)).astype(np.float64))
        mag = abs(u)
        p_val = np.sum((mag**ppow).astype(np.float64))
        g_p = np.sign(u) * mag**(ppow - 1)
        scd = self.g[:, ys, xs]
        tg = self.params['tv'] * g_tv
        pg = self.params['p'] * g_p
        grad = scd + tg
        grad = grad + pg
        self._grad_tile = np.ascontiguousarray(grad, F32)
        if getattr(self, '_no_update', False):
            sq = lambda z: np.sum(z.astype(np.float64)**2)
            img = [tv_val, p_val, sq(scd), sq(tg), sq(pg), sq(grad)]
            s2 = [] if self._s2_reduced else list(self._s2)
            self._p3 = np.asarray(img + s2, F32)
            return torch.from_numpy(self._p3)
        self.items1 += 1
        self.items2 += 1
        self.m = F32(0.9) * self.m + F32(1 - 0.9) * grad
        self.v = F32(0.999) * self.v + F32(1 - 0.999) * grad**2
        m_hat = self.m / F32(1 - 0.9**self.items1)
        v_hat = self.v / F32(1 - 0.999**self.items2)
        nxt = self.x[self.cur ^ 1]
        nxt[0][:, ys, xs] = xt - self.step_size * m_hat / (np.sqrt(v_hat) + 1e-8)
        sq = lambda z: np.sum(z.astype(np.float64)**2)
        img = [tv_val, p_val, sq(scd), sq(tg), sq(pg), sq(grad)]
        s2 = [] if self._s2_reduced else list(self._s2)
        self._p3 = np.asarray(img + s2, F32)
        return torch.from_numpy(self._p3)

    def finish_trace(self):
        """Same layout as the engine: 6 per active layer (c_loss,c_grad,s_loss,s_grad,d_loss,d_grad) + 8."""
        vals = []
        loss = F32(0)
        s2 = self._s2 if self._s2_reduced else self._p3[6:]
        k = 0
        for name, cw, sw, dw, c_on, s_on, d_on in self.active:
            n = self._n_global(name)
            sums, _ = self.red[name]
            v6 = [0.0] * 6
            if c_on:
                cn = self.norms['c', name]
                v6[0] = cw * F32(sums[0] / n) / cn
                v6[1] = abs(cw) * np.sqrt(F32(sums[1] / n)) / cn
                loss += v6[0]
            if s_on:
                sn = self.norms['s', name]
                D = self.D[name]
                v6[2] = sw * np.mean(D**2) / sn
                v6[3] = abs(sw / sn) * np.sqrt(F32(s2[k] / n))
                loss += v6[2]
                k += 1
            if d_on:
                dn = self.norms['d', name]
                v6[4] = -dw * F32(sums[2] / n) / dn
                v6[5] = abs(dw) * np.sqrt(F32(sums[3] / n)) / dn
                loss += v6[4]
            vals += v6
        n3 = 3.0 * self.grid.gH * self.grid.gW
        im = self._p3
        scd_loss = loss
        t_loss = F32(self.params['tv']) * im[0]
        p_loss = F32(self.params['p']) * (im[1] / F32(self.params['p_power']))
        total = scd_loss + t_loss + p_loss
        vals += [scd_loss, t_loss, p_loss, np.sqrt(F32(im[2] / n3)), np.sqrt(F32(im[3] / n3)),
                 np.sqrt(F32(im[4] / n3)), total, np.sqrt(F32(im[5] / n3))]
        return np.asarray(vals, np.float64)
