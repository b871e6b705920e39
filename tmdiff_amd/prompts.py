"""The five sensor paragraphs WavBEST conditions on (data, reference GeneralModel/Hyper_unet_general.py:574-585).

The embedding of a paragraph depends on its exact text, so the strings are kept verbatim -- including the
reference's quirk that the "WV2" entry is the GaoFen-2 paragraph with 0.5 m / 2.0 m resolutions (:581-582).
"""

_TAIL_GF = (" These physical properties enable accurate Earth observation, supporting applications in urban planning, "
            "environmental monitoring, disaster management, and land use analysis.")
_GF_BANDS = ("in four spectral bands: blue (450-520 nm), green (520-590 nm), red (630-690 nm), and near-infrared "
             "(770-890 nm).")


def _gaofen(pan_res, ms_res):
    return ("The GaoFen-2 satellite captures high-resolution images with notable physical properties. Its panchromatic "
            f"sensor acquires images at a {pan_res}-meter resolution, delivering clear and detailed grayscale visuals. "
            f"The multispectral sensor captures images at a {ms_res}-meter resolution " + _GF_BANDS + _TAIL_GF)


PROMPT_TEXT = {
    "QB": ("The QuickBird satellite captures high-resolution images with notable physical properties. Its panchromatic "
           "sensor acquires images at a 0.61-meter resolution, providing crisp and detailed grayscale visuals. The "
           "multispectral sensor captures images at a 2.44-meter resolution in four spectral bands: blue (450-520 nm), "
           "green (520-600 nm), red (630-690 nm), and near-infrared (760-900 nm). These physical properties enable "
           "accurate Earth observation, supporting applications in environmental monitoring, land use planning, urban "
           "mapping, and disaster management."),
    "WV3": ("The WorldView-3 satellite captures high-resolution images with exceptional physical properties. Its "
            "panchromatic sensor acquires images at a 31 cm resolution, delivering sharp and detailed grayscale visuals. "
            "The multispectral sensor captures images at a 1.24 m resolution in eight spectral bands: coastal "
            "(400-450 nm), blue (450-510 nm), green (510-580 nm), yellow (585-625 nm), red (630-690 nm), red edge "
            "(705-745 nm), near-infrared 1 (770-895 nm), and near-infrared 2 (860-1,040 nm). Additionally, WorldView-3 "
            "features a shortwave infrared (SWIR) sensor with 3.7 m resolution in eight bands (1,195-1,385 nm, "
            "1,560-1,660 nm, 2,045-2,110 nm, etc.). These physical properties enable advanced Earth observation, "
            "supporting applications in environmental monitoring, land use planning, urban mapping, and disaster "
            "response."),
    "GF2": _gaofen("1.0", "4.0"),
    "WV2": _gaofen("0.5", "2.0"),
    "WV4": ("The WorldView-4 satellite captures high-resolution images with remarkable physical properties. Its "
            "panchromatic sensor acquires images at a 31 cm resolution, providing sharp, detailed grayscale visuals. The "
            "multispectral sensor captures images at a 1.24 m resolution in four spectral bands: blue (450-510 nm), green "
            "(510-580 nm), red (630-690 nm), and near-infrared (770-895 nm). These physical properties enable precise "
            "Earth observation, facilitating applications in environmental monitoring, land use planning, and disaster "
            "response."),
}
