// oracle/ref_quant_glue.cpp — TEST INFRASTRUCTURE ONLY.
// C entry point around the reference's own convertFlowToImage; appended to the reference lines by
// oracle/Makefile (target `ref`), never compiled on its own.
extern "C" void ref_convert_flow_to_image(const float *flow_x, const float *flow_y, int w, int h, double lower_bound,
                                          double upper_bound, unsigned char *img_x, unsigned char *img_y) {
    const Mat fx{h, w, (unsigned char *)flow_x, (size_t)w * 4}, fy{h, w, (unsigned char *)flow_y, (size_t)w * 4};
    Mat ix{h, w, img_x, (size_t)w}, iy{h, w, img_y, (size_t)w};
    convertFlowToImage(fx, fy, ix, iy, lower_bound, upper_bound);
}
