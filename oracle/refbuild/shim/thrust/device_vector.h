// TEST INFRASTRUCTURE ONLY. Empty stand-in: thrust is only used by the host function SimpleKNN::knn
// (simple_knn.cu:193-220), which build_ref.py cuts away; ref_knn.cpp restates that host sequence with std::.
#pragma once
