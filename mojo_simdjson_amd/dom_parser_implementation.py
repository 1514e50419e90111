"""Host-side mirror of the reference's parser facade for the stage-1 path.

Reference: ``src/mojo_simdjson/include/generic/dom_parser_implementation.mojo:15-89``
(``DomParserImplementation``: fields ``:16-27``, ``stage1`` x3 ``:59-69``,
``allocate`` ``:85-89``).  The reference is Mojo and this image has no Mojo
toolchain, so the host side above the C ABI is written in Python with the same
names, argument meaning and error behaviour; the one call it replaces,
``JsonStructuralIndexer.index[128](buffer, self)`` (``:69``), goes through
``msj_stage1`` in ``libmsj_stage1.so`` (HIP kernels on gfx950).  The Mojo shim a
maintainer would add is in INTEGRATION.md.

Only stage 1 is in scope: ``stage2`` (the serial tape builder,
``dom_parser_implementation.mojo:71-83``) is deliberately not provided.
"""
import ctypes

import numpy as np

from . import _lib, errors


class DomParserImplementation:
    def __init__(self):
        # dom_parser_implementation.mojo:29-39
        self.buf = None  # bytes-like passed to stage 1 (kept alive for the consumer)
        self.length = 0
        self.n_structural_indexes = 0
        self.structural_indexes = np.zeros(0, dtype=np.uint32)
        self._storage = np.zeros(0, dtype=np.uint32)  # the list's allocation (its capacity); structural_indexes views it
        self._registered = False
        self.next_structural_index = 0
        self.utf8_verdict = errors.SUCCESS  # extra: the reference's checker is a stub
        self._capacity = 0
        self._max_depth = 100

    def max_depth(self):
        return self._max_depth

    def capacity(self):
        return self._capacity

    # from this size on the list's allocation is pinned where it grows (msj_host_register): the indices then come
    # down by DMA straight into it
    REGISTER_FROM_BYTES = 64 << 20

    def allocate(self, amount):
        """dom_parser_implementation.mojo:85-89 -- ``reserve(amount)`` + ``resize(amount, 0)`` on a list the parser
        keeps: memory is allocated only when a document is larger than any before, and ``resize`` zero-fills only
        the elements it adds.

        The reference sizes the list to exactly ``amount`` slots although the
        callee writes three trailer words after the last index
        (json_structural_indexer.mojo:167-173); the replacement ABI requires
        ``len + 3`` (include/msj_stage1.h), so three extra slots are reserved.
        """
        want = amount + 3
        if want > self._storage.size:  # reserve(): the list moves
            self._unregister()
            self._storage = np.zeros(want, dtype=np.uint32)
            if self._storage.nbytes >= self.REGISTER_FROM_BYTES:
                lib = _lib.load()
                self._registered = lib.msj_host_register(None, self._storage.ctypes.data, self._storage.nbytes) == 0
        elif want > self.structural_indexes.size:  # resize() upwards inside the capacity: the added elements are 0
            self._storage[self.structural_indexes.size:want] = 0
        self.structural_indexes = self._storage[:want]
        self._capacity = amount

    def _unregister(self):
        if self._registered:
            _lib.load().msj_host_unregister(None, self._storage.ctypes.data)
            self._registered = False

    def __del__(self):
        try:
            self._unregister()
        except Exception:  # interpreter shutdown
            pass

    def stage1(self, buffer, flags=0):
        """``stage1(String | StringSlice | Span[UInt8]) -> ErrorType`` (:59-69)."""
        if isinstance(buffer, str):
            buffer = buffer.encode("utf-8")  # String.as_bytes(), :59-63
        data = np.frombuffer(buffer, dtype=np.uint8)
        n_bytes = int(data.size)
        self.allocate(n_bytes)  # :66
        self.buf = buffer       # :67
        self.length = n_bytes   # :68
        lib = _lib.load()
        n = ctypes.c_uint64(self.n_structural_indexes)
        utf8 = ctypes.c_int32(0)
        rc = lib.msj_stage1(
            data.ctypes.data if n_bytes else None,
            n_bytes,
            self.structural_indexes.ctypes.data,
            self.structural_indexes.size,
            ctypes.byref(n),
            ctypes.byref(utf8),
            flags,
        )
        if rc < 0:
            raise RuntimeError(
                f"libmsj_stage1.so failed with {rc} "
                "(-2 = no HIP device: this package has no CPU fallback)"
            )
        if rc in (errors.SUCCESS, errors.EMPTY, errors.UTF8_ERROR) and n_bytes:
            # json_structural_indexer.mojo:160-174 (not reached on 14/15)
            self.n_structural_indexes = int(n.value)
            self.next_structural_index = 0
        self.utf8_verdict = int(utf8.value)
        return rc

    def stage2(self):
        raise NotImplementedError(
            "stage 2 (tape builder) is outside this repo's scope; see DESIGN.md")
