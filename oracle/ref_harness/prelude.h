// prelude.h — pre-included (-include) when compiling the reference's own
// translation units with g++/libstdc++ (TEST INFRASTRUCTURE, authoring
// container only). ExternalLibrary/scene.h:47-53 defines function-like
// min/max macros; objFunctions.cpp then calls std::max(...), which those
// macros would mangle. Declaring the macros object-like first makes scene.h's
// #ifndef skip its own definition while min(n,numChild) (scene.h:460) still
// resolves to std::min. No reference file is modified.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <ctime>
using std::max;
using std::min;
#define min min
#define max max
