"""nerficg_amd/_lib.py -- loads libnerficg_hip.so (the C-ABI HIP library) with ctypes.

The prototypes are parsed from include/nerficg_hip.h, so the Python bindings cannot drift from the header.
There is NO CPU fallback: if the library is missing or a GPU op is called without a GPU, this fails loudly.
"""
from __future__ import annotations

import ctypes
import os
import re
from functools import lru_cache
from pathlib import Path

PKG = Path(__file__).resolve().parent
HEADER = PKG.parent / 'include' / 'nerficg_hip.h'
LIB_PATH = Path(os.environ['NRC_LIB_PATH']) if os.environ.get('NRC_LIB_PATH') else PKG / 'lib' / 'libnerficg_hip.so'  # override: A/B runs of two builds

ERRORS = {0: 'NRC_OK', -1: 'NRC_ERR_INVALID', -2: 'NRC_ERR_LAUNCH', -3: 'NRC_ERR_UNSUPPORTED'}

_SCALARS = {
    'int': ctypes.c_int, 'int32_t': ctypes.c_int32, 'uint32_t': ctypes.c_uint32, 'int64_t': ctypes.c_int64,
    'uint64_t': ctypes.c_uint64, 'uint8_t': ctypes.c_uint8, 'float': ctypes.c_float, 'double': ctypes.c_double, 'nrc_stream_t': ctypes.c_void_p,
}


class NativeLibraryError(ImportError):
    pass


def parse_header(path: Path = HEADER) -> dict[str, tuple[str, list[tuple[str, str]]]]:
    """Returns {symbol: (return_type, [(ctype_string, arg_name), ...])} for every prototype in the header."""
    text = re.sub(r'/\*.*?\*/', ' ', path.read_text(), flags=re.S)
    text = re.sub(r'//[^\n]*', ' ', text)
    protos: dict[str, tuple[str, list[tuple[str, str]]]] = {}
    for m in re.finditer(r'\b(int|int64_t|const\s+char\s*\*)\s+(nrc_\w+)\s*\(([^;{]*?)\)\s*;', text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        parsed = []
        args = ' '.join(args.split())
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                mm = re.match(r'(.*?)(\w+)$', a)
                parsed.append((mm.group(1).strip(), mm.group(2)))
        protos[name] = (' '.join(ret.split()), parsed)
    return protos


def header_abi_version(path: Path = HEADER) -> int:
    """NRC_ABI_VERSION of the header the bindings are built from."""
    m = re.search(r'#define\s+NRC_ABI_VERSION\s+(\d+)', path.read_text())
    if m is None:
        raise NativeLibraryError(f'{path} does not define NRC_ABI_VERSION')
    return int(m.group(1))


def _ctype(type_str: str):
    if '*' in type_str:
        return ctypes.c_void_p
    base = type_str.replace('const', '').strip()
    return _SCALARS[base]


@lru_cache(maxsize=1)
def load() -> ctypes.CDLL:
    # torch ships its own libamdhip64: it must be in the process BEFORE our library is loaded so that both resolve to ONE HIP
    # runtime (loading ours first pulls /opt/rocm's copy and the kernels then launch on a runtime that has no device context:
    # "no ROCm-capable device is detected")
    import torch  # noqa: F401
    if not LIB_PATH.exists():
        raise NativeLibraryError(
            f'{LIB_PATH} is missing: build the HIP extension first (python -m nerficg_amd.build). '
            'nerficg_amd has no CPU fallback.')
    lib = ctypes.CDLL(str(LIB_PATH))
    for name, (ret, args) in parse_header().items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NativeLibraryError(f'{LIB_PATH} does not export {name} declared in {HEADER.name}') from e
        fn.argtypes = [_ctype(t) for t, _ in args]
        fn.restype = ctypes.c_char_p if 'char' in ret else (ctypes.c_int64 if ret == 'int64_t' else ctypes.c_int)
    want, got = header_abi_version(), lib.nrc_abi_version()
    if got != want:   # a stale .so under newer bindings (or the reverse): fail here, not in an out-of-bounds write
        raise NativeLibraryError(f'{LIB_PATH} implements ABI version {got}, {HEADER.name} declares {want}: rebuild (python -m nerficg_amd.build)')
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        detail = ''
        if status == -2:
            detail = f' ({load().nrc_last_error().decode()})'
        raise RuntimeError(f'{what} failed: {ERRORS.get(status, status)}{detail}')


def ptr(t):
    """Device/host pointer of a torch tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_of(t) -> ctypes.c_void_p:
    """torch's current stream on the tensor's device as the raw handle (the launch stream of every entry point).  Through torch's C accessor: the Python
    route -- torch.cuda.current_stream(dev).cuda_stream builds a Stream object -- cost ~6 us a call, eight calls per training iteration."""
    import torch
    raw = getattr(torch._C, '_cuda_getCurrentRawStream', None)
    if raw is not None:
        idx = t.device.index
        return ctypes.c_void_p(raw(idx if idx is not None else torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def check_input(t, name: str, dtype=None) -> None:
    """CHECK_INPUT of the reference (csrc/include/utils.h:4-6): CUDA(HIP) tensor + contiguous, RuntimeError otherwise."""
    if not t.is_cuda:
        raise RuntimeError(f'{name} must be a CUDA tensor')
    if not t.is_contiguous():
        raise RuntimeError(f'{name} must be contiguous')
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f'{name} must have dtype {dtype}, got {t.dtype}')


class HostMailbox:
    """Pinned, device-mapped, coherent host memory that a kernel of the library writes and the host polls (include/nerficg_hip.h:
    nrc_host_mailbox_alloc).  An entry point that takes (count_mailbox, mailbox_ticket) stores two counts and the ticket, the ticket last;
    `wait(ticket)` spins on it -- no event, no device-to-host copy, no stream wait behind the call, which on this runtime cost tens of
    microseconds of idle GPU per frame.  One call in flight per mailbox.  `for_device(dev)` returns the process-wide mailbox of a device,
    or None when the runtime cannot map one (or after one failed to answer: callers then read the device counters)."""

    TIMEOUT_S = 2.0
    _per_device: dict = {}

    def __init__(self, device) -> None:
        import torch
        box = ctypes.c_void_p()
        with torch.cuda.device(device):
            check(load().nrc_host_mailbox_alloc(ctypes.byref(box)), 'host_mailbox_alloc')
        self.ptr = box
        self._seen = (ctypes.c_int64 * 3).from_address(box.value)
        self._ticket = 0
        # one call in flight per mailbox: a caller holds the lock from next_ticket() until wait() has returned (threads that render on the same device)
        import threading
        self.lock = threading.RLock()

    @classmethod
    def for_device(cls, device):
        key = str(device)
        if key not in cls._per_device:
            try:
                cls._per_device[key] = cls(device)
            except RuntimeError:
                cls._per_device[key] = None
        return cls._per_device[key]

    @classmethod
    def retire(cls, device) -> None:
        cls._per_device[str(device)] = None

    def next_ticket(self) -> int:
        self._ticket += 1
        return self._ticket

    def counts(self, ticket: int, device):
        """wait(ticket); when the spin times out (seconds of queued GPU work in front of the posting kernel, say) the caller's stream is
        synchronised and the mailbox looked at once more -- only a mailbox that is STILL silent then is broken: None, and it is retired."""
        got = self.wait(ticket)
        if got is None:
            import torch
            torch.cuda.current_stream(device).synchronize()
            if self._seen[2] == ticket:
                got = int(self._seen[0]), int(self._seen[1])
            else:
                type(self).retire(device)
        return got

    def wait(self, ticket: int):
        """(first, second) count once the kernel that was given `ticket` has stored them; None after TIMEOUT_S of spinning."""
        import time
        seen = self._seen
        spins, deadline = 0, None
        while seen[2] != ticket:
            spins += 1
            if (spins & 0xfff) == 0:   # every few hundred microseconds: look at the clock
                now = time.monotonic()
                deadline = deadline or now + self.TIMEOUT_S
                if now > deadline:
                    return None
        return int(seen[0]), int(seen[1])


class stage_timer:
    """Context manager around the library's stage timer (include/nerficg_hip.h group 12): HIP events on the launch stream behind every kernel
    of the multi-kernel entry points, recorded by the library itself.  After the block, `.stages` is the list of (kernel name, ms) in launch
    order and `.by_name()` the per-name (total ms, launches).  Measurement only; eager calls only (nothing is recorded inside a capture)."""

    def __init__(self, capacity: int = 1 << 16) -> None:
        self.capacity = int(capacity)
        self.stages: list[tuple[str, float]] = []

    def __enter__(self):
        check(load().nrc_stage_timer_begin(self.capacity), 'stage_timer_begin')
        return self

    def __exit__(self, *exc):
        names = ctypes.create_string_buffer(32 * self.capacity)
        ms = (ctypes.c_float * self.capacity)()
        n = ctypes.c_int32(0)
        check(load().nrc_stage_timer_end(self.capacity, ctypes.cast(names, ctypes.c_void_p), ctypes.cast(ms, ctypes.c_void_p),
                                         ctypes.cast(ctypes.pointer(n), ctypes.c_void_p)), 'stage_timer_end')
        raw = names.raw
        self.stages = [(raw[32 * i:32 * i + 32].split(b'\0', 1)[0].decode(), float(ms[i])) for i in range(n.value)]
        return False

    def by_name(self) -> dict[str, tuple[float, int]]:
        out: dict[str, tuple[float, int]] = {}
        for name, t in self.stages:
            tot, cnt = out.get(name, (0.0, 0))
            out[name] = (tot + t, cnt + 1)
        return out
