/* camera.h -- forwarding header: scene programs written for ndt include "../camera.h" (or reach it
 * through "../scene.h"); the whole host API of this repository lives in ndt_host_api.h. */
#include "ndt_host_api.h"
