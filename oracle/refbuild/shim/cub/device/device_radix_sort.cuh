// TEST INFRASTRUCTURE ONLY. Empty stand-in: the CUB calls live in the host
// launchers that are cut away; the driver uses std::stable_sort / a serial scan.
#pragma once
