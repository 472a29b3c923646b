"""A limb-sharded key-switch plan backed by the CPU oracle (TEST INFRASTRUCTURE): the same interface as
fhe_reliability_gpu_amd.dist.ShardedKeySwitch (begin / inner / finish, gather buffers g1 / g2) on CPU int64 tensors,
so dist.sharded_keyswitch's sequencing and the shard layout can run over gloo without a GPU.  The row maps are written
out again here from the layout's definition (not shared with the C++ plan): the two-ranks-on-one-GPU test compares both."""
import numpy as np


class OracleShardPlan:
    def __init__(self, qs, logn, L, K, dnum, group=None):
        import torch

        from fhe_reliability_gpu_amd.dist import _group_info, ks_layout
        from oracle import cport as O
        self.O, self.qs, self.logn, self.L, self.K, self.dnum, self.group = O, [int(q) for q in qs], logn, L, K, dnum, group
        self.N = 1 << logn
        self.alpha = -(-L // dnum)
        self.world, self.rank = _group_info(group)
        self.lays = [ks_layout(L, K, self.world, r) for r in range(self.world)]
        self.lay = self.lays[self.rank]
        self.cmax, self.smax = self.lay["cmax"], self.lay["smax"]
        self.rows1, self.rows2 = self.cmax, 2 * self.smax
        self.g1 = torch.zeros((self.world * self.cmax, self.N), dtype=torch.int64)
        self.g2 = torch.zeros((self.world * 2 * self.smax, self.N), dtype=torch.int64)
        self.rps = np.stack([O.root_powers(q, logn) for q in self.qs])
        self.bc = torch.zeros((3, self.N), dtype=torch.int64)
        self.last_owner = next(r for r, l in enumerate(self.lays) if l["clo"] <= L - 1 < l["clo"] + l["cn"])
        self.owns_last = self.last_owner == self.rank
        self.rs_rows = max(0, min(self.lay["clo"] + self.lay["cn"], L - 1) - self.lay["clo"])
        lay = self.lay
        self.own = list(range(lay["clo"], lay["clo"] + lay["cn"])) + list(range(lay["slo"], lay["slo"] + lay["sn"]))

    @staticmethod
    def _np(x):
        return np.ascontiguousarray(x.numpy()).view(np.uint64)

    @staticmethod
    def _t(a):
        import torch
        return torch.from_numpy(np.ascontiguousarray(a).view(np.int64).copy())

    def _row1(self, l):
        for r, lay in enumerate(self.lays):
            if lay["clo"] <= l < lay["clo"] + lay["cn"]:
                return r * self.cmax + l - lay["clo"]
        raise AssertionError("ciphertext limb without an owner")

    def _row2(self, k, h):
        for r, lay in enumerate(self.lays):
            if lay["slo"] <= self.L + k < lay["slo"] + lay["sn"]:
                return (r * 2 + h) * self.smax + self.L + k - lay["slo"]
        raise AssertionError("special limb without an owner")

    def begin(self, c_local):
        lay = self.lay
        for j in range(lay["cn"]):
            l = lay["clo"] + j
            self.g1[self.rank * self.cmax + j] = self._t(self.O.nwt_inverse(self._np(c_local[j]), self.qs[l], self.rps[l]))

    def inner(self, c_local, evk_local):
        O, lay, N = self.O, self.lay, self.N
        own = self.own
        acc = np.zeros((2, len(own), N), dtype=np.uint64)
        for d in range(self.dnum):
            lo, hi = d * self.alpha, min(self.L, (d + 1) * self.alpha)
            digit = np.stack([self._np(self.g1[self._row1(l)]) for l in range(lo, hi)])
            for jj, tl in enumerate(own):
                if lo <= tl < hi:
                    x = self._np(c_local[tl - lay["clo"]])
                else:
                    x = O.nwt_forward(O.baseconv_exact(digit, self.qs[lo:hi], [self.qs[tl]])[0], self.qs[tl], self.rps[tl])
                for h in range(2):
                    acc[h, jj] = O.modmul_acc(acc[h, jj], x, self._np(evk_local[d, h, jj]), self.qs[tl])
        self.acc = acc
        for h in range(2):
            for kk in range(lay["sn"]):
                tl = lay["slo"] + kk
                self.g2[(self.rank * 2 + h) * self.smax + kk] = self._t(O.nwt_inverse(acc[h, lay["cn"] + kk], self.qs[tl], self.rps[tl]))

    def finish(self, add0=None, add1=None):
        import torch
        O, lay, N, L, K = self.O, self.lay, self.N, self.L, self.K
        P = self.qs[L:L + K]
        outs = []
        for h, add in enumerate((add0, add1)):
            out = torch.zeros((lay["cn"], N), dtype=torch.int64)
            if lay["cn"]:
                tP = np.stack([self._np(self.g2[self._row2(k, h)]) for k in range(K)])
                for j in range(lay["cn"]):
                    tl = lay["clo"] + j
                    q = self.qs[tl]
                    cn = O.nwt_forward(O.baseconv_exact(tP, P, [q])[0], q, self.rps[tl])
                    pm = 1
                    for pk in P:
                        pm = pm * (pk % q) % q
                    d = (self.acc[h, j] + (np.uint64(q) - cn)) % np.uint64(q)
                    v = O.modmul(d, np.full(N, pow(pm, -1, q), dtype=np.uint64), q)
                    if add is not None:
                        v = (v + self._np(add[j]) % np.uint64(q)) % np.uint64(q)
                    out[j] = self._t(v)
            outs.append(out)
        return outs[0], outs[1]

    # ---- hoisted rotations (tests of dist.sharded_rotate_hoisted): the same phases as the C ABI's fhe_rotate_hoisted_shard_*
    def _sigma(self, x, k, tl):
        from oracle.keyswitch_ref import galois_coeff
        q = self.qs[tl]
        return self.O.nwt_forward(galois_coeff(self.O.nwt_inverse(x, q, self.rps[tl]), k, q), q, self.rps[tl])

    def prepare_galois_key(self, gk_local, galois_elt):
        import torch
        kinv = pow(int(galois_elt), -1, 2 * self.N)
        out = torch.zeros_like(gk_local)
        for d in range(self.dnum):
            for h in range(2):
                for jj, tl in enumerate(self.own):
                    out[d, h, jj] = self._t(self._sigma(self._np(gk_local[d, h, jj]), kinv, tl))
        return out

    def hoisted_begin(self, c1_local):
        self.begin(c1_local)

    def hoisted_extend(self):
        O, lay = self.O, self.lay
        self.ext = []                                   # [digit][owned row] NTT form; None where the digit's own limb takes c1 itself
        for d in range(self.dnum):
            lo, hi = d * self.alpha, min(self.L, (d + 1) * self.alpha)
            digit = np.stack([self._np(self.g1[self._row1(l)]) for l in range(lo, hi)])
            row = []
            for tl in self.own:
                row.append(None if lo <= tl < hi else O.nwt_forward(O.baseconv_exact(digit, self.qs[lo:hi], [self.qs[tl]])[0], self.qs[tl], self.rps[tl]))
            self.ext.append(row)

    def hoisted_inner(self, c1_local, pk_local, galois_elt):
        O, lay, N = self.O, self.lay, self.N
        acc = np.zeros((2, len(self.own), N), dtype=np.uint64)
        for d in range(self.dnum):
            for jj, tl in enumerate(self.own):
                x = self.ext[d][jj] if self.ext[d][jj] is not None else self._np(c1_local[tl - lay["clo"]])
                for h in range(2):
                    acc[h, jj] = O.modmul_acc(acc[h, jj], x, self._np(pk_local[d, h, jj]), self.qs[tl])
        # sigma of the sums: the un-rotated frame ends here
        self.acc = np.stack([np.stack([self._sigma(acc[h, jj], galois_elt, tl) for jj, tl in enumerate(self.own)]) for h in range(2)])
        for h in range(2):
            for kk in range(lay["sn"]):
                tl = lay["slo"] + kk
                self.g2[(self.rank * 2 + h) * self.smax + kk] = self._t(O.nwt_inverse(self.acc[h, lay["cn"] + kk], self.qs[tl], self.rps[tl]))

    def hoisted_finish(self, c0_local, galois_elt):
        lay = self.lay
        sig0 = None
        if lay["cn"]:
            import torch
            sig0 = torch.stack([self._t(self._sigma(self._np(c0_local[j]), galois_elt, lay["clo"] + j)) for j in range(lay["cn"])])
        return self.finish(sig0, None)

    # ---- homomorphic multiply on the owned rows (tests of dist.sharded_hmult)
    def tensor(self, a0, a1, b0, b1):
        import torch
        lay = self.lay
        d = [torch.zeros_like(a0) for _ in range(3)]
        for j in range(lay["cn"]):
            q = self.qs[lay["clo"] + j]
            x0, x1, y0, y1 = (self._np(t[j]) for t in (a0, a1, b0, b1))
            d[0][j] = self._t(self.O.modmul(x0, y0, q))
            d[1][j] = self._t(self.O.modmul_acc(self.O.modmul(x0, y1, q), x1, y0, q))
            d[2][j] = self._t(self.O.modmul(x1, y1, q))
        return tuple(d)

    def rescale_begin(self, parts_local):
        if not self.owns_last:
            return
        L = self.L
        row = L - 1 - self.lay["clo"]
        for p in range(parts_local.shape[0]):
            self.bc[p] = self._t(self.O.nwt_inverse(self._np(parts_local[p, row]), self.qs[L - 1], self.rps[L - 1]))

    def rescale_finish(self, parts_local):
        import torch
        O, lay, N, L = self.O, self.lay, self.N, self.L
        ql = self.qs[L - 1]
        out = torch.zeros((parts_local.shape[0], self.rs_rows, N), dtype=torch.int64)
        for p in range(parts_local.shape[0]):
            y = self._np(self.bc[p])
            for j in range(self.rs_rows):
                tl = lay["clo"] + j
                q = self.qs[tl]
                dn = O.nwt_forward(y % np.uint64(q), q, self.rps[tl])
                c = self._np(parts_local[p, j])
                diff = (c % np.uint64(q) + (np.uint64(q) - dn)) % np.uint64(q)
                out[p, j] = self._t(O.modmul(diff, np.full(N, pow(ql % q, -1, q), dtype=np.uint64), q))
        return out


    # ---- hmult with the broadcast between the conversion and the last transform (dist.sharded_hmult's fused flow,
    # fhe_hmult_shard_finish_begin / _end): stated here with the plain formulas -- mod-down of the owned rows, the owner's last limb
    # to coefficient form for the broadcast, then the rescale of the rows below it -- the protocol (who holds what when) is what the
    # gloo tests exercise; the engine's one-transform arithmetic is checked against these words on the GPU
    fused_rescale = True

    def hm_finish_begin(self, add0, add1):
        import torch
        c0, c1 = self.finish(add0, add1)
        self._hm_parts = torch.stack([c0, c1])
        self.rescale_begin(self._hm_parts[:, :, :])

    def hm_finish_end(self, add0, add1):
        return self.rescale_finish(self._hm_parts)
