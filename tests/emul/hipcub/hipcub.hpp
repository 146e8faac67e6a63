// test infrastructure: the two hipCUB algorithms the host code calls are emulated in ../hip_emul.h
#include "../hip_emul.h"
