// test infrastructure: the emulation stand-in of <hip/hip_runtime.h> (see ../hip_emul.h)
#include "../hip_emul.h"
