#include "emu_dyn.h"
RTB_EMU_DYN_DISPATCH(emu_dyn_r6, RTB_EMU_DYN(16))
