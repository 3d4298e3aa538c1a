// ORACLE — TEST INFRASTRUCTURE ONLY (see DBoW3/src/DBoW3.h: cv::Mat stand-in).
#pragma once
#include "DBoW3/src/DBoW3.h"
