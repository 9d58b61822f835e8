// test stand-in (integration/shim/README.md).  Only what hip::Scene::toPod() reads of the render
// parameters, as plain members without defaults - the test host fills in every one of them.
#pragma once
struct RenderParams {
  int seed, maxDepth, samplesPerPixel;
  int firstBounceUSamples, firstBounceVSamples;
  int width, height;
  bool preview;
};
