#include "emu_ik_wave.h"
RTB_EMU_IK_DISPATCH(emu_ik_wave_mid, RTB_EMU_IK(8) RTB_EMU_IK(9) RTB_EMU_IK(10) RTB_EMU_IK(11) RTB_EMU_IK(12))
