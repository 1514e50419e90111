"""Error codes of the stage-1 path.

Mirror of the reference's ``src/mojo_simdjson/errors.mojo:2-36`` (same names,
same integer values; ``ErrorType = Int`` there, plain ``int`` here).
"""
SUCCESS = 0
CAPACITY = 1
MEMALLOC = 2
TAPE_ERROR = 3
DEPTH_ERROR = 4
STRING_ERROR = 5
T_ATOM_ERROR = 6
F_ATOM_ERROR = 7
N_ATOM_ERROR = 8
NUMBER_ERROR = 9
BIGINT_ERROR = 10
UTF8_ERROR = 11
UNINITIALIZED = 12
EMPTY = 13
UNESCAPED_CHARS = 14
UNCLOSED_STRING = 15
UNSUPPORTED_ARCHITECTURE = 16
UNEXPECTED_ERROR = 24

# library-level failures of libmsj_stage1.so (include/msj_stage1.h)
ERR_BAD_ARGUMENT = -1
ERR_NO_DEVICE = -2
ERR_HIP = -3

ErrorType = int
