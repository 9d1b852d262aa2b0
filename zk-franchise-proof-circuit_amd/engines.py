"""The NTT and G1 MSM engines on their own -- the ffjavascript calls under snarkjs' groth16.prove (`Fr.fft`, `Fr.ifft`,
`G1.multiExpAffine`; ts_inputs/src/example.ts:358) over device buffers.  Thin ctypes wrappers of include/zkcensus.h
zkc_ntt_dev / zkc_g1_mul_batch_dev / zkc_msm_g1_*; used by SURVEY.md 8(d) config 5 (ii) (tools/stress.py) and its parity tests."""
import ctypes

R_MONT = 1 << 256
G1_GENERATOR = (1).to_bytes(32, 'little') + (2).to_bytes(32, 'little')


def fft(ctx, d_src_ptr, d_dst_ptr, logn, nvec=1):
    """Forward NTT of nvec vectors of 2^logn Montgomery-form Fr elements (natural order in and out); d_src != d_dst."""
    ctx._check(ctx._lib.zkc_ntt_dev(ctx._h, d_src_ptr, d_dst_ptr, logn, nvec, 0))


def ifft(ctx, d_src_ptr, d_dst_ptr, logn, nvec=1):
    ctx._check(ctx._lib.zkc_ntt_dev(ctx._h, d_src_ptr, d_dst_ptr, logn, nvec, 1))


def g1_mul_batch(ctx, base64, d_scalars_ptr, n, d_out_ptr):
    """d_out[i] = k_i * base; scalars 32 B standard form, points affine standard form (64 B)."""
    ctx._check(ctx._lib.zkc_g1_mul_batch_dev(ctx._h, bytes(base64), d_scalars_ptr, n, d_out_ptr))


class G1Bases:
    """n fixed G1 bases resident as pre-shifted window tables; multiExpAffine(scalars) = sum s_i P_i."""

    def __init__(self, ctx, d_bases_ptr, n):
        self.ctx, self.n = ctx, n
        h = ctypes.c_void_p()
        ctx._check(ctx._lib.zkc_msm_g1_load_dev(ctx._h, d_bases_ptr, n, ctypes.byref(h)))
        self._h = h

    def multiExpAffine(self, d_scalars_ptr):
        out = ctypes.create_string_buffer(64)
        self.ctx._check(self.ctx._lib.zkc_msm_g1_dev(self._h, d_scalars_ptr, out))
        return out.raw

    def close(self):
        if getattr(self, '_h', None):
            self.ctx._lib.zkc_msm_g1_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
