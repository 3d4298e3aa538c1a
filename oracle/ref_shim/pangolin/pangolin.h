// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-in for <pangolin/pangolin.h>: frontend/DSOViewer.h (pulled in by FullSystem.h) declares two
// pangolin::GlBuffer members; the viewer is never constructed in the pin (FullSystem::viewer stays nullptr).
#pragma once
namespace pangolin { struct GlBuffer {}; }
