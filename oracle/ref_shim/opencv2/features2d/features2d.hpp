// ORACLE - TEST INFRASTRUCTURE ONLY.  Empty stand-in (FullSystem.cc includes it; the pin never calls OpenCV).
#pragma once
#include <opencv2/opencv.hpp>
