// oracle/ref_png_glue.cpp — TEST INFRASTRUCTURE ONLY.
// C entry point around the reference's own convertFlowToPngImage; appended to the reference lines by
// oracle/Makefile (target `ref`), never compiled on its own.
extern "C" void ref_convert_flow_to_png_image(const float *flow_x, const float *flow_y, int w, int h,
                                              unsigned char *img_bgr) {
    const Mat fx(h, w, CV_32FC1, (void *)flow_x, (size_t)w * 4), fy(h, w, CV_32FC1, (void *)flow_y, (size_t)w * 4);
    Mat out(h, w, CV_8UC3, img_bgr, (size_t)w * 3);
    convertFlowToPngImage(fx, fy, out);
}
