// TEST INFRASTRUCTURE ONLY. Empty stand-in (the reference includes but never uses cg::reduce).
#pragma once
