// Scenes of 9 .. 32 persons: csrc/grecon.hip compiled a second time with the person arrays of the scene description sized for 32, in
// namespace glamr::grecon_wide -- the stage kernel's lite-arena and workspace instances for several persons and the two entry points
// glamr_grecon_run_stage / glamr_grecon_workspace_bytes forward to (as *_wide_).  The reference loops over the persons of a scene in
// Python without a limit (/root/reference/global_recon/models/global_recon_model.py:154,432,511).
#define GLAMR_GRECON_WIDE 1
#define GLAMR_MAX_PERSONS 32
#define GLAMR_GRECON_NS grecon_wide
#include "grecon.hip"
