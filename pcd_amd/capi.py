"""ctypes binding of libpcdhip.so (include/pcdhip.h).  numpy uint64 arrays in, numpy arrays out.

Mirrors the C-ABI one to one; the class/method names follow the upstream functions they stand in
for (`multi_scalar_mul`, `fft`/`ifft`/`coset_fft`/`coset_ifft`, `witness_map`, `create_proof`).
No fallback: a missing library or a missing GPU raises PcdHipError."""
import ctypes as C
import os

import numpy as np

from .build import library_path

MNT4_298, MNT6_298, MNT4_753, MNT6_753 = 0, 1, 2, 3
G1, G2 = 1, 2
FIELD_LIMBS = [5, 5, 12, 12]
CURVE_FQ = [0, 1, 2, 3]
CURVE_FR = [1, 0, 3, 2]
CURVE_G2_DEG = [2, 3, 2, 3]


class PcdHipError(RuntimeError):
    pass


class PrevTicketError(PcdHipError):
    """PCDHIP_E_PREV_TICKET: an EARLIER MSM of the slot had an unreduced scalar; `ticket` (when not None) is the NEW submission, which was
    enqueued all the same and must be waited for / released like any other"""

    def __init__(self, msg, ticket=None):
        super().__init__(msg)
        self.ticket = ticket


class Csr(C.Structure):
    _fields_ = [("num_rows", C.c_uint64), ("row_ptr", C.c_void_p), ("col", C.c_void_p), ("coeff", C.c_void_p)]


class G16PkHost(C.Structure):
    _fields_ = [("curve_id", C.c_uint32), ("_pad", C.c_uint32), ("num_vars", C.c_uint64), ("num_inputs", C.c_uint64),
                ("domain_size", C.c_uint64),
                ("alpha_g1", C.c_void_p), ("beta_g1", C.c_void_p), ("delta_g1", C.c_void_p),
                ("beta_g2", C.c_void_p), ("delta_g2", C.c_void_p),
                ("a_query", C.c_void_p), ("a_inf", C.c_void_p),
                ("b_g1_query", C.c_void_p), ("b_g1_inf", C.c_void_p),
                ("b_g2_query", C.c_void_p), ("b_g2_inf", C.c_void_p),
                ("h_query", C.c_void_p), ("h_inf", C.c_void_p), ("h_len", C.c_uint64),
                ("l_query", C.c_void_p), ("l_inf", C.c_void_p), ("l_len", C.c_uint64)]


_LIB = None

EXPORTS = [
    "pcdhip_strerror", "pcdhip_device_count", "pcdhip_init", "pcdhip_init_devices", "pcdhip_ctx_devices", "pcdhip_destroy", "pcdhip_sync", "pcdhip_host_alloc", "pcdhip_host_free", "pcdhip_last_hip_error",
    "pcdhip_field_limbs", "pcdhip_curve_base_field", "pcdhip_curve_scalar_field", "pcdhip_point_limbs",
    "pcdhip_buf_upload", "pcdhip_buf_alloc", "pcdhip_buf_download", "pcdhip_buf_free",
    "pcdhip_bases_upload", "pcdhip_bases_free", "pcdhip_msm", "pcdhip_msm_dev", "pcdhip_msm_config", "pcdhip_msm_submit", "pcdhip_msm_collect", "pcdhip_msm_submit_partial", "pcdhip_msm_ticket_wait", "pcdhip_msm_ticket_status", "pcdhip_bases_info", "pcdhip_stream_wait",
    "pcdhip_set_precompute", "pcdhip_set_precompute_budget", "pcdhip_get_precompute_budget", "pcdhip_msm_set_sort", "pcdhip_msm_set_accumulate", "pcdhip_msm_profile", "pcdhip_msm_last_timings", "pcdhip_msm_last_plan", "pcdhip_mad_rate", "pcdhip_points_sum", "pcdhip_msm_dev_partial", "pcdhip_points_sum_dev", "pcdhip_to_affine",
    "pcdhip_fft", "pcdhip_fft_dev", "pcdhip_fft_general", "pcdhip_fft_seq", "pcdhip_domain_size", "pcdhip_fft_last_timings", "pcdhip_groth16_witness_map",
    "pcdhip_g16_pk_upload", "pcdhip_g16_pk_free", "pcdhip_g16_pk_set_r1cs", "pcdhip_g16_pk_info", "pcdhip_g16_pk_memory", "pcdhip_g16_witness_map_resident", "pcdhip_groth16_prove", "pcdhip_groth16_last_timings", "pcdhip_groth16_set_assembly", "pcdhip_groth16_set_sparse_window", "pcdhip_groth16_last_plan", "pcdhip_groth16_set_witness_split", "pcdhip_groth16_set_schedule", "pcdhip_set_lane_reserve", "pcdhip_fixed_base_mul", "pcdhip_groth16_setup",
    "pcdhip_serialized_size", "pcdhip_serialize_points", "pcdhip_deserialize_points", "pcdhip_deserialize_points_unchecked", "pcdhip_proof_serialized_size", "pcdhip_proof_serialize",
    "pcdhip_proof_deserialize", "pcdhip_vk_serialized_size", "pcdhip_vk_serialize", "pcdhip_vk_deserialize",
    "pcdhip_process_vk", "pcdhip_pvk_free", "pcdhip_groth16_verify_prepared", "pcdhip_groth16_verify_batch_rlc",
    "pcdhip_multi_pairing", "pcdhip_pairing_set_mode", "pcdhip_groth16_verify", "pcdhip_groth16_verify_batch", "pcdhip_timer_start", "pcdhip_timer_stop",
]


def lib():
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise PcdHipError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback)")
        _LIB = C.CDLL(path)
        _LIB.pcdhip_strerror.restype = C.c_char_p
        _LIB.pcdhip_last_hip_error.restype = C.c_char_p
        _LIB.pcdhip_destroy.restype = None
        _LIB.pcdhip_domain_size.restype = C.c_size_t
        for name in ("pcdhip_serialized_size", "pcdhip_proof_serialized_size", "pcdhip_vk_serialized_size"):
            getattr(_LIB, name).restype = C.c_size_t
        for name in ("pcdhip_buf_free", "pcdhip_bases_free", "pcdhip_g16_pk_free", "pcdhip_host_free", "pcdhip_pvk_free"):
            getattr(_LIB, name).restype = None
    return _LIB


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def point_limbs(curve, group):
    return 2 * (1 if group == G1 else CURVE_G2_DEG[curve]) * FIELD_LIMBS[CURVE_FQ[curve]]


class Context:
    """One device + one HIP stream (pcdhip_ctx); with `devices=[...]` a multi-device context (pcdhip_init_devices) that shards
    bases, keys, MSMs and proofs by point range over the listed devices."""

    def __init__(self, device=0, devices=None):
        self._ctx = C.c_void_p()
        if devices is not None:
            ids = (C.c_int * len(devices))(*[int(d) for d in devices])
            rc = lib().pcdhip_init_devices(ids, len(devices), C.byref(self._ctx))
            device = devices[0]
        else:
            rc = lib().pcdhip_init(int(device), C.byref(self._ctx))
        if rc != 0:
            raise PcdHipError(f"pcdhip_init(device={device}, devices={devices}) failed: {lib().pcdhip_strerror(rc).decode()}")
        self.device = device
        self.devices = list(devices) if devices is not None else [device]

    def close(self):
        if self._ctx:
            lib().pcdhip_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            detail = lib().pcdhip_last_hip_error(self._ctx).decode()
            raise PcdHipError(f"{lib().pcdhip_strerror(rc).decode()} (rc={rc}) {detail}")

    def sync(self):
        self._check(lib().pcdhip_sync(self._ctx))

    # ---- device vectors
    def buf_upload(self, field, arr):
        arr = _u64(arr)
        n = arr.shape[0]
        h = C.c_void_p()
        self._check(lib().pcdhip_buf_upload(self._ctx, field, _p(arr), C.c_size_t(n), C.byref(h)))
        return DeviceBuf(self, h, field, n)

    # ---- MSM
    def bases_upload(self, curve, group, xy, inf=None):
        xy = _u64(xy)
        n = xy.shape[0]
        infp = np.ascontiguousarray(inf, dtype=np.uint8) if inf is not None else None
        h = C.c_void_p()
        self._check(lib().pcdhip_bases_upload(self._ctx, curve, group, _p(xy), _p(infp), C.c_size_t(n), C.byref(h)))
        return Bases(self, h, curve, group, n)

    def msm(self, bases, scalars, offset=0, n=None):
        """multi_scalar_mul(bases[offset:offset+n], scalars) -> Jacobian X||Y||Z (uint64 limbs)."""
        out = np.zeros(3 * point_limbs(bases.curve, bases.group) // 2, dtype=np.uint64)
        if isinstance(scalars, DeviceBuf):
            n = scalars.n if n is None else n
            self._check(lib().pcdhip_msm_dev(self._ctx, bases._h, C.c_size_t(offset), scalars._h, C.c_size_t(0), C.c_size_t(n), _p(out)))
        else:
            scalars = _u64(scalars)
            n = scalars.shape[0] if n is None else n
            self._check(lib().pcdhip_msm(self._ctx, bases._h, C.c_size_t(offset), _p(scalars), C.c_size_t(n), _p(out)))
        return out

    def msm_submit(self, bases, scalars, offset=0, n=None, scalar_offset=0):
        """enqueue an MSM over resident scalars on a side stream -> ticket (at most four outstanding); see msm_collect"""
        n = scalars.n - scalar_offset if n is None else n
        t = C.c_int(-1)
        self._check(lib().pcdhip_msm_submit(self._ctx, bases._h, C.c_size_t(offset), scalars._h, C.c_size_t(scalar_offset), C.c_size_t(n), C.byref(t)))
        return (t.value, bases.curve, bases.group)

    def msm_collect(self, ticket):
        t, curve, group = ticket
        out = np.zeros(3 * point_limbs(curve, group) // 2, dtype=np.uint64)
        self._check(lib().pcdhip_msm_collect(self._ctx, t, _p(out)))
        return out

    def msm_submit_partial(self, bases, scalars, out_slots_device_ptr, slot_stride_bytes, offset=0, n=None):
        """msm_submit with the Jacobian partial left in slot `ticket` of the caller's device buffer (four slots, `slot_stride_bytes`
        apart) -> ticket number; release it with msm_ticket_wait(ticket, stream), which makes `stream` wait for the MSM."""
        n = scalars.n if n is None else n
        t = C.c_int(-1)
        rc = lib().pcdhip_msm_submit_partial(self._ctx, bases._h, C.c_size_t(offset), scalars._h, C.c_size_t(0), C.c_size_t(n),
                                             C.c_void_p(out_slots_device_ptr), C.c_size_t(slot_stride_bytes), C.byref(t))
        if rc == -6:   # PCDHIP_E_PREV_TICKET: this submission is in flight; an earlier one of the slot was wrong
            raise PrevTicketError(lib().pcdhip_strerror(rc).decode(), t.value)
        self._check(rc)
        return t.value

    def msm_ticket_wait(self, ticket, other_stream):
        self._check(lib().pcdhip_msm_ticket_wait(self._ctx, int(ticket), C.c_void_p(other_stream)))

    def msm_ticket_status(self, slot):
        """waits for the last released MSM of `slot`; raises PrevTicketError when one of its scalars was not reduced"""
        rc = lib().pcdhip_msm_ticket_status(self._ctx, int(slot))
        if rc == -6:
            raise PrevTicketError(lib().pcdhip_strerror(rc).decode())
        self._check(rc)

    def msm_partial_to_device(self, bases, scalars, out_device_ptr, offset=0, n=None):
        """The MSM of a shard with its Jacobian result left at `out_device_ptr` (device memory of the caller, e.g. a torch
        tensor's data_ptr(): the send buffer of the all-gather).  Asynchronous: call sync() before another stream reads it."""
        n = scalars.n if n is None else n
        self._check(lib().pcdhip_msm_dev_partial(self._ctx, bases._h, C.c_size_t(offset), scalars._h, C.c_size_t(0), C.c_size_t(n),
                                                 C.c_void_p(out_device_ptr)))

    def points_sum_device(self, curve, group, xyz_device_ptr, n):
        """Sum of n Jacobian points that sit in device memory (the receive buffer of the all-gather) -> host limbs."""
        out = np.zeros(3 * point_limbs(curve, group) // 2, dtype=np.uint64)
        self._check(lib().pcdhip_points_sum_dev(self._ctx, curve, group, C.c_void_p(xyz_device_ptr), C.c_size_t(n), _p(out)))
        return out

    def bases_info(self, bases, n=0):
        """(window bits c, scalar windows W, resident window-shifted copies) of an MSM of n pairs over `bases`."""
        c, w, k = C.c_int(), C.c_int(), C.c_int()
        self._check(lib().pcdhip_bases_info(bases._h, C.c_size_t(n), C.byref(c), C.byref(w), C.byref(k)))
        return c.value, w.value, k.value

    def stream_wait(self, other_stream, direction):
        """direction 0: this context's stream waits for `other_stream` (a raw hipStream_t); 1: the other way round."""
        self._check(lib().pcdhip_stream_wait(self._ctx, C.c_void_p(other_stream), int(direction)))

    def msm_config(self, window_bits=0, chunk=0):
        self._check(lib().pcdhip_msm_config(self._ctx, window_bits, chunk))

    def set_precompute(self, mode):
        """-1 full (default), 0 none, k >= 2 copies; applies to bases uploaded afterwards."""
        self._check(lib().pcdhip_set_precompute(self._ctx, int(mode)))

    def set_precompute_budget(self, bytes_per_vector):
        """bytes one base vector uploaded afterwards may occupy with its window-shifted copies (0: no bound); fewer copies otherwise."""
        self._check(lib().pcdhip_set_precompute_budget(self._ctx, C.c_size_t(int(bytes_per_vector))))

    def get_precompute_budget(self):
        v = C.c_size_t(0)
        self._check(lib().pcdhip_get_precompute_budget(self._ctx, C.byref(v)))
        return v.value

    def msm_set_sort(self, mode):
        """0: LDS partition sort (default); 1: single-pass binning with on-device fallback; 2: two-pass counting sort."""
        self._check(lib().pcdhip_msm_set_sort(self._ctx, int(mode)))

    def msm_set_accumulate(self, mode, chunk=0, min_pairs=0):
        """0 by size, 1 running sums, 2 pair tree of affine additions (753-bit G1); chunk / min_pairs: tuning and test knobs"""
        self._check(lib().pcdhip_msm_set_accumulate(self._ctx, int(mode), int(chunk), int(min_pairs)))

    def msm_profile(self, on=True):
        self._check(lib().pcdhip_msm_profile(self._ctx, int(on)))

    def msm_last_timings(self):
        out = (C.c_float * 8)()
        self._check(lib().pcdhip_msm_last_timings(self._ctx, out))
        return dict(zip(["digits", "scan", "scatter", "accumulate", "fixup", "tail", "horner", "total"], list(out)))

    def msm_last_plan(self):
        """(entries of the sorted list, entries per lane the device chose) of the last profiled MSM"""
        out = (C.c_uint32 * 2)()
        self._check(lib().pcdhip_msm_last_plan(self._ctx, out))
        return int(out[0]), int(out[1])

    def mad_rate(self):
        """v_mad_u64_u32 lane-operations per second of this device, measured now (pcdhip_mad_rate)"""
        out = C.c_double(0.0)
        self._check(lib().pcdhip_mad_rate(self._ctx, C.byref(out)))
        return float(out.value)

    def points_sum(self, curve, group, xyz):
        xyz = _u64(xyz).reshape(-1, 3 * point_limbs(curve, group) // 2)
        out = np.zeros(xyz.shape[1], dtype=np.uint64)
        self._check(lib().pcdhip_points_sum(self._ctx, curve, group, _p(xyz), C.c_size_t(xyz.shape[0]), _p(out)))
        return out

    def to_affine(self, curve, group, xyz):
        xyz = _u64(xyz).reshape(-1, 3 * point_limbs(curve, group) // 2)
        n = xyz.shape[0]
        xy = np.zeros((n, point_limbs(curve, group)), dtype=np.uint64)
        inf = np.zeros(n, dtype=np.uint8)
        self._check(lib().pcdhip_to_affine(self._ctx, curve, group, _p(xyz), C.c_size_t(n), _p(xy), _p(inf)))
        return xy, inf

    # ---- FFT
    def fft(self, field, data, inverse=False, coset=False):
        if isinstance(data, DeviceBuf):
            log_n = data.n.bit_length() - 1
            self._check(lib().pcdhip_fft_dev(self._ctx, data._h, log_n, int(inverse), int(coset)))
            return data
        data = _u64(data).copy()
        n = data.shape[0]
        log_n = n.bit_length() - 1
        assert 1 << log_n == n
        self._check(lib().pcdhip_fft(self._ctx, field, _p(data), log_n, int(inverse), int(coset)))
        return data

    def fft_general(self, field, data, inverse=False, coset=False):
        """GeneralEvaluationDomain transform on n = 2^a q^b elements (mixed radix for the help fields)."""
        data = _u64(data).copy()
        self._check(lib().pcdhip_fft_general(self._ctx, field, _p(data), C.c_size_t(data.shape[0]), int(inverse), int(coset)))
        return data

    def fft_seq(self, field, data, ops):
        """the transforms `ops` = [(inverse, coset), ..] applied one after the other to one host vector, one trip over PCIe"""
        data = _u64(data).copy()
        codes = (C.c_int * len(ops))(*[int(bool(i)) | (int(bool(c)) << 1) for i, c in ops])
        self._check(lib().pcdhip_fft_seq(self._ctx, field, _p(data), C.c_size_t(data.shape[0]), codes, len(ops)))
        return data

    def fft_last_timings(self):
        out = (C.c_float * 8)()
        k = lib().pcdhip_fft_last_timings(self._ctx, out)
        return list(out)[:max(k, 0)]

    # ---- Groth16
    @staticmethod
    def _csr(rp, col, coeff):
        s = Csr()
        s.num_rows = len(rp) - 1
        s.row_ptr, s.col, s.coeff = rp.ctypes.data, col.ctypes.data, coeff.ctypes.data
        return s

    def witness_map(self, field, r1cs):
        """R1CSToQAP::witness_map: r1cs has rp_/col_/coeff_{a,b,c}, z, num_inputs (any object with those numpy arrays)."""
        A = self._csr(r1cs.rp_a, r1cs.col_a, r1cs.coeff_a)
        B = self._csr(r1cs.rp_b, r1cs.col_b, r1cs.coeff_b)
        Cm = self._csr(r1cs.rp_c, r1cs.col_c, r1cs.coeff_c)
        n = lib().pcdhip_domain_size(field, C.c_size_t(r1cs.num_constraints + r1cs.num_inputs))
        h = np.zeros((n, FIELD_LIMBS[field]), dtype=np.uint64)
        self._check(lib().pcdhip_groth16_witness_map(self._ctx, field, C.byref(A), C.byref(B), C.byref(Cm), _p(r1cs.z),
                                                     C.c_size_t(r1cs.num_vars), C.c_size_t(r1cs.num_inputs), _p(h)))
        return h

    def g16_pk_upload(self, host_struct, curve):
        h = C.c_void_p()
        self._check(lib().pcdhip_g16_pk_upload(self._ctx, C.byref(host_struct), C.byref(h)))
        return G16Pk(self, h, curve)

    def g16_pk_set_r1cs(self, pk, r1cs):
        A = self._csr(r1cs.rp_a, r1cs.col_a, r1cs.coeff_a)
        B = self._csr(r1cs.rp_b, r1cs.col_b, r1cs.coeff_b)
        Cm = self._csr(r1cs.rp_c, r1cs.col_c, r1cs.coeff_c)
        self._check(lib().pcdhip_g16_pk_set_r1cs(self._ctx, pk._h, C.byref(A), C.byref(B), C.byref(Cm)))

    def witness_map_resident(self, pk, r1cs, want_h=True):
        """witness map alone over the matrices resident with pk -> (h or None, {"spmv", "transforms", "total"} device ms)."""
        n = lib().pcdhip_domain_size(CURVE_FR[pk.curve], C.c_size_t(r1cs.num_constraints + r1cs.num_inputs))
        h = np.zeros((n, FIELD_LIMBS[CURVE_FR[pk.curve]]), dtype=np.uint64) if want_h else None
        ms = (C.c_float * 3)()
        self._check(lib().pcdhip_g16_witness_map_resident(self._ctx, pk._h, _p(r1cs.z), _p(h), ms))
        return h, dict(zip(["spmv", "transforms", "total"], list(ms)))

    def groth16_prove(self, pk, r1cs, r_mont, s_mont, resident_r1cs=False):
        """create_proof after synthesis -> (proof A||B||C affine limbs, inf flags[3]).  With resident_r1cs the
        matrices set by g16_pk_set_r1cs are used and only z is uploaded."""
        w1, w2 = point_limbs(pk.curve, G1), point_limbs(pk.curve, G2)
        proof = np.zeros(2 * w1 + w2, dtype=np.uint64)
        inf = np.zeros(3, dtype=np.uint8)
        if resident_r1cs:
            self._check(lib().pcdhip_groth16_prove(self._ctx, pk._h, None, None, None, _p(r1cs.z),
                                                   _p(_u64(r_mont)), _p(_u64(s_mont)), _p(proof), _p(inf)))
            return proof, inf
        A = self._csr(r1cs.rp_a, r1cs.col_a, r1cs.coeff_a)
        B = self._csr(r1cs.rp_b, r1cs.col_b, r1cs.coeff_b)
        Cm = self._csr(r1cs.rp_c, r1cs.col_c, r1cs.coeff_c)
        self._check(lib().pcdhip_groth16_prove(self._ctx, pk._h, C.byref(A), C.byref(B), C.byref(Cm), _p(r1cs.z),
                                               _p(_u64(r_mont)), _p(_u64(s_mont)), _p(proof), _p(inf)))
        return proof, inf

    def g16_pk_info(self, pk):
        """{query: (window bits, windows)} of a resident key's a / b_g1 / b_g2 / l / h queries"""
        c, w = (C.c_int * 5)(), (C.c_int * 5)()
        self._check(lib().pcdhip_g16_pk_info(pk._h, c, w))
        return {k: (int(c[i]), int(w[i])) for i, k in enumerate(("a", "b_g1", "b_g2", "l", "h"))}

    def g16_pk_memory(self, pk):
        """device bytes of a resident key's base vectors: {"ordinary", "sparse_window", "copies", "sparse_copies"}"""
        out = (C.c_uint64 * 4)()
        self._check(lib().pcdhip_g16_pk_memory(pk._h, out))
        return {"ordinary": int(out[0]), "sparse_window": int(out[1]), "copies": int(out[2]), "sparse_copies": int(out[3])}

    def groth16_set_sparse_window(self, bits):
        """0 off (default), -1 automatic, 6..22 forced window bits of the sparse-assignment copies (takes effect at the next key upload)"""
        self._check(lib().pcdhip_groth16_set_sparse_window(self._ctx, int(bits)))

    def groth16_last_plan(self):
        """(the last proof ran its assignment MSMs on the sparse-window copies, its count of general scalars)"""
        out = (C.c_uint32 * 2)()
        self._check(lib().pcdhip_groth16_last_plan(self._ctx, out))
        return bool(out[0]), int(out[1])

    def groth16_set_assembly(self, mode):
        """0: automatic (default); 1: s*A, r*B_1 folded into two extra MSMs; 2: chained one-lane products overlapping the other MSMs."""
        self._check(lib().pcdhip_groth16_set_assembly(self._ctx, int(mode)))

    def groth16_set_schedule(self, mode):
        """0 (default): the assignment MSMs first, the witness map under them; 1: the witness map first, then all five MSMs at once;
        2: the witness map first, then the accumulate lane (round 5; measured slower, see include/pcdhip.h)"""
        self._check(lib().pcdhip_groth16_set_schedule(self._ctx, int(mode)))

    def set_lane_reserve(self, cus):
        """compute units the accumulate lane leaves to the other streams (-1: default 8 / PCDHIP_LANE_RESERVE; 0: no CU mask)"""
        self._check(lib().pcdhip_set_lane_reserve(self._ctx, int(cus)))

    def groth16_set_witness_split(self, on):
        """multi-device contexts of >= 3 devices: the witness map's a / b / c chains on devices 0 / 1 / 2 (default) or all on device 0"""
        self._check(lib().pcdhip_groth16_set_witness_split(self._ctx, int(bool(on))))

    def groth16_last_timings(self):
        out = (C.c_float * 8)()
        self._check(lib().pcdhip_groth16_last_timings(self._ctx, out))
        return dict(zip(["witness_map", "msm_h", "msm_l", "msm_a", "msm_b_g1", "msm_b_g2", "assembly", "total"], list(out)))

    # ---- key generation
    def fixed_base_mul(self, curve, group, base_xy, scalars_canonical):
        """FixedBaseMSM::multi_scalar_mul + batch normalisation: (n affine points, n flags) = scalars[i] * base."""
        sc = _u64(scalars_canonical).reshape(-1, FIELD_LIMBS[CURVE_FR[curve]])
        n = sc.shape[0]
        out = np.zeros((n, point_limbs(curve, group)), dtype=np.uint64)
        inf = np.zeros(n, dtype=np.uint8)
        self._check(lib().pcdhip_fixed_base_mul(self._ctx, curve, group, _p(_u64(base_xy)), _p(sc), C.c_size_t(n), _p(out), _p(inf)))
        return out, inf

    def groth16_setup(self, curve, r1cs, g1_xy, g2_xy, toxic_mont):
        """ark-groth16 generate_parameters after synthesis (circuit_specific_setup, src/ec_cycle_pcd/mod.rs:69,78);
        toxic_mont = (alpha, beta, gamma, delta, tau).  Returns a dict of arrays named like the key's fields."""
        w1, w2 = point_limbs(curve, G1), point_limbs(curve, G2)
        m, ni = r1cs.num_vars, r1cs.num_inputs
        n = lib().pcdhip_domain_size(CURVE_FR[curve], C.c_size_t(r1cs.num_constraints + ni))
        z64 = lambda *sh: np.zeros(sh, dtype=np.uint64)
        z8 = lambda k: np.zeros(k, dtype=np.uint8)
        K = dict(alpha_g1=z64(w1), beta_g1=z64(w1), delta_g1=z64(w1), beta_g2=z64(w2), gamma_g2=z64(w2), delta_g2=z64(w2),
                 a_query=z64(m, w1), a_inf=z8(m), b_g1_query=z64(m, w1), b_g1_inf=z8(m), b_g2_query=z64(m, w2), b_g2_inf=z8(m),
                 h_query=z64(max(n - 1, 0), w1), h_inf=z8(max(n - 1, 0)), l_query=z64(m - ni, w1), l_inf=z8(m - ni),
                 gamma_abc_g1=z64(ni, w1), gamma_abc_inf=z8(ni))
        out = G16SetupOut()
        for name, _ in G16SetupOut._fields_:
            if name != "domain_size":
                setattr(out, name, K[name].ctypes.data if K[name].size else None)
        A = self._csr(r1cs.rp_a, r1cs.col_a, r1cs.coeff_a)
        B = self._csr(r1cs.rp_b, r1cs.col_b, r1cs.coeff_b)
        Cm = self._csr(r1cs.rp_c, r1cs.col_c, r1cs.coeff_c)
        self._check(lib().pcdhip_groth16_setup(self._ctx, curve, C.byref(A), C.byref(B), C.byref(Cm), C.c_size_t(m), C.c_size_t(ni),
                                               _p(_u64(g1_xy)), _p(_u64(g2_xy)), _p(_u64(toxic_mont)), C.byref(out)))
        assert out.domain_size == n
        K["domain_size"] = n
        return K

    # ---- pairing
    def multi_pairing(self, curve, g1_xy, g2_xy, g1_inf=None, g2_inf=None):
        """product_of_pairings: final_exponentiation(prod miller_loop(P_i, Q_i)) -> GT limbs (tower order)."""
        g1 = _u64(g1_xy).reshape(-1, point_limbs(curve, G1))
        g2 = _u64(g2_xy).reshape(-1, point_limbs(curve, G2))
        n = g1.shape[0]
        out = np.zeros(2 * CURVE_G2_DEG[curve] * FIELD_LIMBS[CURVE_FQ[curve]], dtype=np.uint64)
        i1 = np.ascontiguousarray(g1_inf, dtype=np.uint8) if g1_inf is not None else None
        i2 = np.ascontiguousarray(g2_inf, dtype=np.uint8) if g2_inf is not None else None
        self._check(lib().pcdhip_multi_pairing(self._ctx, curve, _p(g1), _p(i1), _p(g2), _p(i2), C.c_size_t(n), _p(out)))
        return out

    def pairing_set_mode(self, mode):
        """0: one wave per pairing for batches up to 4096 pairs (default); 1: always one lane per pairing."""
        self._check(lib().pcdhip_pairing_set_mode(self._ctx, int(mode)))

    def groth16_verify(self, curve, alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1, public_inputs_canonical, proof,
                       gamma_abc_inf=None, proof_inf=None):
        """Groth16::verify (reference call site src/ec_cycle_pcd/mod.rs:239)."""
        abc = _u64(gamma_abc_g1).reshape(-1, point_limbs(curve, G1))
        ok = C.c_int(0)
        pi = _u64(public_inputs_canonical)
        gi = np.ascontiguousarray(gamma_abc_inf, dtype=np.uint8) if gamma_abc_inf is not None else None
        pinf = np.ascontiguousarray(proof_inf, dtype=np.uint8) if proof_inf is not None else None
        self._check(lib().pcdhip_groth16_verify(self._ctx, curve, _p(_u64(alpha_g1)), _p(_u64(beta_g2)), _p(_u64(gamma_g2)),
                                                _p(_u64(delta_g2)), _p(abc), _p(gi), C.c_size_t(abc.shape[0]), _p(pi),
                                                _p(_u64(proof)), _p(pinf), C.byref(ok)))
        return ok.value == 1

    def groth16_verify_batch(self, curve, alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1, public_inputs_canonical, proofs,
                             gamma_abc_inf=None, proofs_inf=None):
        """n Groth16 verifications under one key (the inputs of a merge node) with all Miller loops in one launch -> bool array."""
        abc = _u64(gamma_abc_g1).reshape(-1, point_limbs(curve, G1))
        ni = abc.shape[0]
        pr = _u64(proofs).reshape(-1, 2 * point_limbs(curve, G1) + point_limbs(curve, G2))
        k = pr.shape[0]
        pi = _u64(public_inputs_canonical).reshape(k, (ni - 1) * FIELD_LIMBS[CURVE_FR[curve]]) if ni > 1 else None
        ok = (C.c_int * max(k, 1))()
        gi = np.ascontiguousarray(gamma_abc_inf, dtype=np.uint8) if gamma_abc_inf is not None else None
        pinf = np.ascontiguousarray(proofs_inf, dtype=np.uint8) if proofs_inf is not None else None
        self._check(lib().pcdhip_groth16_verify_batch(self._ctx, curve, _p(_u64(alpha_g1)), _p(_u64(beta_g2)), _p(_u64(gamma_g2)),
                                                      _p(_u64(delta_g2)), _p(abc), _p(gi), C.c_size_t(ni), C.c_size_t(k), _p(pi), _p(pr),
                                                      _p(pinf), ok))
        return np.array([ok[i] == 1 for i in range(k)])

    def process_vk(self, curve, alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1, gamma_abc_inf=None):
        """SNARK::process_vk: e(alpha, beta) cached, gamma / delta negated, gamma_abc resident."""
        abc = _u64(gamma_abc_g1).reshape(-1, point_limbs(curve, G1))
        gi = np.ascontiguousarray(gamma_abc_inf, dtype=np.uint8) if gamma_abc_inf is not None else None
        h = C.c_void_p()
        self._check(lib().pcdhip_process_vk(self._ctx, curve, _p(_u64(alpha_g1)), _p(_u64(beta_g2)), _p(_u64(gamma_g2)), _p(_u64(delta_g2)),
                                            _p(abc), _p(gi), C.c_size_t(abc.shape[0]), C.byref(h)))
        return Pvk(self, h, curve, abc.shape[0])

    def _proof_batch(self, pvk, public_inputs_canonical, proofs, proofs_inf):
        pr = _u64(proofs).reshape(-1, 2 * point_limbs(pvk.curve, G1) + point_limbs(pvk.curve, G2))
        k = pr.shape[0]
        pi = _u64(public_inputs_canonical).reshape(k, (pvk.num_inputs - 1) * FIELD_LIMBS[CURVE_FR[pvk.curve]]) if pvk.num_inputs > 1 else None
        pinf = np.ascontiguousarray(proofs_inf, dtype=np.uint8) if proofs_inf is not None else None
        return pr, k, pi, pinf

    def groth16_verify_prepared(self, pvk, public_inputs_canonical, proofs, proofs_inf=None):
        """verify_with_processed_vk for n proofs -> bool array (3 Miller loops + 1 final exponentiation per proof)."""
        pr, k, pi, pinf = self._proof_batch(pvk, public_inputs_canonical, proofs, proofs_inf)
        ok = (C.c_int * max(k, 1))()
        self._check(lib().pcdhip_groth16_verify_prepared(self._ctx, pvk._h, C.c_size_t(k), _p(pi), _p(pr), _p(pinf), ok))
        return np.array([ok[i] == 1 for i in range(k)])

    def groth16_verify_batch_rlc(self, pvk, public_inputs_canonical, proofs, rho, proofs_inf=None):
        """all n proofs with ONE shared final exponentiation; rho: (n, 2) uint64 non-zero 128-bit challenges -> bool."""
        pr, k, pi, pinf = self._proof_batch(pvk, public_inputs_canonical, proofs, proofs_inf)
        rho = _u64(rho).reshape(k, 2)
        ok = C.c_int(0)
        self._check(lib().pcdhip_groth16_verify_batch_rlc(self._ctx, pvk._h, C.c_size_t(k), _p(pi), _p(pr), _p(pinf), _p(rho), C.byref(ok)))
        return ok.value == 1

    # ---- timing
    def timer_start(self):
        self._check(lib().pcdhip_timer_start(self._ctx))

    def timer_stop(self):
        ms = C.c_float()
        self._check(lib().pcdhip_timer_stop(self._ctx, C.byref(ms)))
        return ms.value


# ---- wire format (host-side: no context, no GPU)
def _wire_check(rc):
    if rc != 0:
        raise PcdHipError(f"{lib().pcdhip_strerror(rc).decode()} (rc={rc})")


def serialize_points(curve, group, xy, inf=None, compressed=True):
    """CanonicalSerialize of n affine points -> bytes."""
    xy = _u64(xy).reshape(-1, point_limbs(curve, group))
    n = xy.shape[0]
    infp = np.ascontiguousarray(inf, dtype=np.uint8) if inf is not None else None
    out = np.zeros(n * lib().pcdhip_serialized_size(curve, group, int(compressed)), dtype=np.uint8)
    _wire_check(lib().pcdhip_serialize_points(curve, group, _p(xy), _p(infp), C.c_size_t(n), int(compressed), _p(out)))
    return out.tobytes()


def deserialize_points(curve, group, data, n, compressed=True, unchecked=False):
    """CanonicalDeserialize of n affine points; unchecked=True skips the G2 subgroup test (`deserialize_unchecked`)."""
    buf = np.frombuffer(data, dtype=np.uint8).copy()
    xy = np.zeros((n, point_limbs(curve, group)), dtype=np.uint64)
    inf = np.zeros(n, dtype=np.uint8)
    fn = lib().pcdhip_deserialize_points_unchecked if unchecked else lib().pcdhip_deserialize_points
    _wire_check(fn(curve, group, _p(buf), C.c_size_t(n), int(compressed), _p(xy), _p(inf)))
    return xy, inf


def proof_serialize(curve, proof, proof_inf=None, compressed=True):
    out = np.zeros(lib().pcdhip_proof_serialized_size(curve, int(compressed)), dtype=np.uint8)
    pinf = np.ascontiguousarray(proof_inf, dtype=np.uint8) if proof_inf is not None else None
    _wire_check(lib().pcdhip_proof_serialize(curve, _p(_u64(proof)), _p(pinf), int(compressed), _p(out)))
    return out.tobytes()


def proof_deserialize(curve, data, compressed=True):
    buf = np.frombuffer(data, dtype=np.uint8).copy()
    proof = np.zeros(2 * point_limbs(curve, G1) + point_limbs(curve, G2), dtype=np.uint64)
    inf = np.zeros(3, dtype=np.uint8)
    _wire_check(lib().pcdhip_proof_deserialize(curve, _p(buf), int(compressed), _p(proof), _p(inf)))
    return proof, inf


def vk_serialize(curve, alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1, gamma_abc_inf=None, compressed=True):
    abc = _u64(gamma_abc_g1).reshape(-1, point_limbs(curve, G1))
    ni = abc.shape[0]
    gi = np.ascontiguousarray(gamma_abc_inf, dtype=np.uint8) if gamma_abc_inf is not None else None
    out = np.zeros(lib().pcdhip_vk_serialized_size(curve, C.c_size_t(ni), int(compressed)), dtype=np.uint8)
    _wire_check(lib().pcdhip_vk_serialize(curve, _p(_u64(alpha_g1)), _p(_u64(beta_g2)), _p(_u64(gamma_g2)), _p(_u64(delta_g2)), _p(abc), _p(gi),
                                          C.c_size_t(ni), int(compressed), _p(out)))
    return out.tobytes()


def vk_deserialize(curve, data, compressed=True, max_inputs=1 << 16):
    buf = np.frombuffer(data, dtype=np.uint8).copy()
    l1, l2 = point_limbs(curve, G1), point_limbs(curve, G2)
    a, b, g, d = (np.zeros(k, dtype=np.uint64) for k in (l1, l2, l2, l2))
    abc = np.zeros((max_inputs, l1), dtype=np.uint64)
    inf = np.zeros(max_inputs, dtype=np.uint8)
    cnt = C.c_size_t(0)
    _wire_check(lib().pcdhip_vk_deserialize(curve, _p(buf), C.c_size_t(len(buf)), int(compressed), _p(a), _p(b), _p(g), _p(d), _p(abc), _p(inf),
                                            C.c_size_t(max_inputs), C.byref(cnt)))
    return dict(alpha_g1=a, beta_g2=b, gamma_g2=g, delta_g2=d, gamma_abc_g1=abc[:cnt.value].copy(), gamma_abc_inf=inf[:cnt.value].copy())


def pinned_like(a):
    """Copy of `a` in page-locked host memory (pcdhip_host_alloc); the array keeps the allocation alive."""
    a = np.ascontiguousarray(a)
    p = C.c_void_p()
    rc = lib().pcdhip_host_alloc(C.c_size_t(max(a.nbytes, 1)), C.byref(p))
    if rc:
        raise PcdHipError(f"pcdhip_host_alloc: {lib().pcdhip_strerror(rc).decode()}")
    buf = (C.c_char * max(a.nbytes, 1)).from_address(p.value)
    out = np.frombuffer(buf, dtype=a.dtype, count=a.size).reshape(a.shape)
    out[...] = a
    _PINNED.append((out, p))   # freed at interpreter exit with the process
    return out


_PINNED = []


class G16SetupOut(C.Structure):
    """Mirror of `pcdhip_g16_setup_out` (include/pcdhip.h)."""
    _fields_ = [(k, C.c_void_p) for k in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "gamma_g2", "delta_g2", "a_query", "a_inf",
                                          "b_g1_query", "b_g1_inf", "b_g2_query", "b_g2_inf", "h_query", "h_inf", "l_query", "l_inf",
                                          "gamma_abc_g1", "gamma_abc_inf")] + [("domain_size", C.c_uint64)]


class DeviceBuf:
    def __init__(self, ctx, h, field, n):
        self.ctx, self._h, self.field, self.n = ctx, h, field, n

    def download(self):
        out = np.zeros((self.n, FIELD_LIMBS[self.field]), dtype=np.uint64)
        self.ctx._check(lib().pcdhip_buf_download(self.ctx._ctx, self._h, _p(out), C.c_size_t(self.n)))
        return out

    def free(self):
        if self._h:
            lib().pcdhip_buf_free(self.ctx._ctx, self._h)
            self._h = None


class Bases:
    def __init__(self, ctx, h, curve, group, n):
        self.ctx, self._h, self.curve, self.group, self.n = ctx, h, curve, group, n

    def free(self):
        if self._h:
            lib().pcdhip_bases_free(self.ctx._ctx, self._h)
            self._h = None


class Pvk:
    def __init__(self, ctx, h, curve, num_inputs):
        self.ctx, self._h, self.curve, self.num_inputs = ctx, h, curve, num_inputs

    def free(self):
        if self._h:
            lib().pcdhip_pvk_free(self.ctx._ctx, self._h)
            self._h = None


class G16Pk:
    def __init__(self, ctx, h, curve):
        self.ctx, self._h, self.curve = ctx, h, curve

    def free(self):
        if self._h:
            lib().pcdhip_g16_pk_free(self.ctx._ctx, self._h)
            self._h = None
